// oracle/ref_shim32.cpp -- TEST INFRASTRUCTURE ONLY.  extern "C" trampolines into the reference's FLOAT build
// (ofpix_t = float, see ofpix_float.h): double planes in / out, converted at the boundary.  Used to measure what float
// storage costs the reference itself, next to the GPU's OFX_F32 mode (tests/test_oracle_vs_ref.py, test_gpu_tvl1.py).
#include <stdexcept>
#include <vector>

#include "tvl1flow.h"
#include "horn_schunck.h"
#include "brox_optic_flow.h"

#include <omp.h>

namespace {
std::vector<float> f32(const double *p, size_t n) { return std::vector<float>(p, p + n); }
void back(const std::vector<float> &v, double *p) { for (size_t i = 0; i < v.size(); i++) p[i] = v[i]; }
}

extern "C" {

void ref32_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }

int ref32_tvl1_multiscale(const double *I0, const double *I1, double *u1, double *u2, int nx, int ny, double tau,
                          double lambda, double theta, int nscales, double zfactor, int warps, double epsilon, int verbose)
{
    const size_t n = (size_t) nx * ny;
    std::vector<float> a = f32(I0, n), b = f32(I1, n), u(n), v(n);
    try {
        Dual_TVL1_optic_flow_multiscale(a.data(), b.data(), u.data(), v.data(), nx, ny, tau, lambda, theta, nscales, zfactor,
                                        warps, epsilon, verbose != 0);
    } catch (const std::runtime_error &) { return 1; }
    back(u, u1); back(v, u2);
    return 0;
}

int ref32_hs_pyramidal(const double *I1, const double *I2, double *uo, double *vo, int nx, int ny, double alpha, int nscales,
                       double zfactor, int warps, double TOL, int maxiter, int verbose)
{
    const size_t n = (size_t) nx * ny;
    std::vector<float> a = f32(I1, n), b = f32(I2, n), u(n), v(n);
    try {
        horn_schunck_pyramidal(a.data(), b.data(), u.data(), v.data(), nx, ny, alpha, nscales, zfactor, warps, TOL, maxiter,
                               verbose != 0);
    } catch (const std::runtime_error &) { return 1; }
    back(u, uo); back(v, vo);
    return 0;
}

} // extern "C"
