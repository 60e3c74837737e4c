// shim_driver.cpp -- a "caller written against the reference headers": it includes the reference's own .h files (found
// with -I/root/reference/src at build time) and calls the reference's own function names; the definitions come from
// include/ofx_reference_shim.hpp + libofx.so instead of the reference's .cpp files.  tests/test_gpu_shim.py runs it.
//
//   shim_driver <in.bin> <out.bin>
// in.bin : int32 nx, ny, frames; then frames * nx * ny doubles (a synthetic sequence)
// out.bin: the doubles every call produced, in call order
#include <cstdio>
#include <vector>

#include "ofx_reference_shim.hpp"

static void put(FILE *f, const std::vector<ofpix_t> &v) { fwrite(v.data(), sizeof(ofpix_t), v.size(), f); }

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) return 2;
    int hdr[3];
    if (fread(hdr, sizeof(int), 3, fi) != 3) return 2;
    const int nx = hdr[0], ny = hdr[1], frames = hdr[2], n = nx * ny;
    std::vector<ofpix_t> I((size_t) n * frames);
    if (fread(I.data(), sizeof(ofpix_t), I.size(), fi) != I.size()) return 2;
    fclose(fi);
    FILE *fo = fopen(argv[2], "wb");
    if (!fo) return 2;
    ofpix_t *I0 = I.data(), *I1 = I.data() + n;
    try {
        std::vector<ofpix_t> a(n), b(n), c(n), u(n), v(n);
        // operators
        divergence(I0, I1, a.data(), nx, ny); put(fo, a);
        forward_gradient(I0, a.data(), b.data(), nx, ny); put(fo, a); put(fo, b);
        centered_gradient(I0, a.data(), b.data(), nx, ny, 1); put(fo, a); put(fo, b);
        Dxx(I0, a.data(), nx, ny, 1); put(fo, a);
        Dyy(I0, a.data(), nx, ny, 1); put(fo, a);
        Dxy(I0, a.data(), nx, ny, 1); put(fo, a);
        a.assign(I0, I0 + n);
        gaussian(a.data(), nx, ny, 0.8); put(fo, a);
        for (int i = 0; i < n; i++) { u[i] = 1.5 + 0.01 * (i % nx); v[i] = -0.75 + 0.02 * (i / nx); }
        bicubic_interpolation_warp(I1, u.data(), v.data(), a.data(), nx, ny, true); put(fo, a);
        a[0] = bicubic_interpolation_at(I0, 3.25, 2.5, nx, ny);
        a[1] = bicubic_interpolation_at(I0, -1.0, 2.5, nx, ny, true);
        fwrite(a.data(), sizeof(ofpix_t), 2, fo);
        int nxx, nyy;
        zoom_size(nx, ny, &nxx, &nyy, 0.5);
        std::vector<ofpix_t> z((size_t) nxx * nyy);
        zoom_out(I0, z.data(), nx, ny, 0.5); put(fo, z);
        zoom_in(z.data(), a.data(), nxx, nyy, nx, ny); put(fo, a);
        image_normalization_2(I0, I1, a.data(), b.data(), n); put(fo, a); put(fo, b);
        ofpix_t mn, mx;
        getminmax(&mn, &mx, I0, n);
        fwrite(&mn, sizeof(ofpix_t), 1, fo);
        fwrite(&mx, sizeof(ofpix_t), 1, fo);
        // solvers, with the reference's own argument lists
        Dual_TVL1_optic_flow_multiscale(I0, I1, u.data(), v.data(), nx, ny, 0.25, 0.15, 0.3, 3, 0.5, 5, 0.01, false);
        put(fo, u); put(fo, v);
        horn_schunck_pyramidal(I0, I1, u.data(), v.data(), nx, ny, 20.0, 3, 0.5, 4, 1e-4, 150, false);
        put(fo, u); put(fo, v);
        brox_optic_flow_spatial(I0, I1, u.data(), v.data(), nx, ny, 50.0, 10.0, 3, 0.5, 1e-4, 1, 4, false);
        put(fo, u); put(fo, v);
        hs(u.data(), v.data(), I0, I1, nx, ny, 25, 15.0);
        put(fo, u); put(fo, v);
        std::vector<ofpix_t> us((size_t) n * (frames - 1)), vs((size_t) n * (frames - 1));
        brox_optic_flow_temporal(I.data(), us.data(), vs.data(), nx, ny, frames, 18.0, 7.0, 2, 0.75, 1e-4, 1, 3, false);
        put(fo, us); put(fo, vs);
        // TV-L1 with occlusions through the reference's own overload (src/tvl1occflow.h), and its stateless helpers
        if (frames >= 3) {
            std::vector<ofpix_t> chi(n);
            ofpix_t *I2 = I.data() + 2 * (size_t) n;
            Dual_TVL1_optic_flow_multiscale(I0, I1, I2, I1, u.data(), v.data(), chi.data(), nx, ny, 0.15, 0.01, 0.15, 0.3, 3, 0.5, 2, 0.01, false);
            put(fo, u); put(fo, v); put(fo, chi);
            a.assign(I0, I0 + n);
            me_median_filtering(a.data(), nx, ny, 3); put(fo, a);
            image_normalization_4(I0, I1, I2, I1, a.data(), b.data(), c.data(), u.data(), n);
            put(fo, a); put(fo, b); put(fo, c); put(fo, u);
        }
        // robust_expo_methods through its own header (src/robust_expo_methods.h), one channel
        robust_expo_methods(I0, I1, u.data(), v.data(), nx, ny, 1, 2, 18.7, 5.0, 0.05, 2, 0.5, 1e-4, 1, 3, false);
        put(fo, u); put(fo, v);
        // the reference's failure mode: an exception with its own text
        try {
            std::vector<ofpix_t> tiny(9, 1.0);
            gaussian(tiny.data(), 3, 3, 0.8);
            fprintf(stderr, "no exception\n");
            return 3;
        } catch (const std::runtime_error &e) {
            if (std::string(e.what()) != "GaussianSmooth: sigma too large") return 3;
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "shim_driver: %s\n", e.what());
        return 1;
    }
    fclose(fo);
    return 0;
}
