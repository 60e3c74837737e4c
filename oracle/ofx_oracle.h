/* oracle/ofx_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C + OpenMP, double storage and arithmetic) of the reference hot path.
 * It is the CHECKER for the HIP product path: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product library (libofx.so) never links,
 * loads or falls back to anything in oracle/.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (OMP_NUM_THREADS=1)
 * against the reference's own compiled sources (oracle/_ref/libofref.so, built by
 * oracle/Makefile from /root/reference/src where they lie) in tests/test_oracle_vs_ref.py, and
 * against golden vectors generated from that build (tests/golden/, generator
 * tests/golden/make_golden.py).  The reference itself ships no tests or fixtures.
 *
 * All arrays are dense row-major nx*ny doubles, index p = i*nx + j.
 */
#ifndef OFX_ORACLE_H
#define OFX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

void orc_set_num_threads(int n);
int  orc_max_threads(void);

/* operators.cpp */
void orc_divergence(const double *v1, const double *v2, double *div, int nx, int ny);
void orc_forward_gradient(const double *f, double *fx, double *fy, int nx, int ny);
void orc_centered_gradient(const double *f, double *dx, double *dy, int nx, int ny);
void orc_dxx(const double *f, double *out, int nx, int ny);
void orc_dyy(const double *f, double *out, int nx, int ny);
void orc_dxy(const double *f, double *out, int nx, int ny);
int  orc_gaussian(double *I, int nx, int ny, double sigma);        /* 1 = "sigma too large" */

/* bicubic_interpolation.cpp */
double orc_bicubic_at(const double *in, double uu, double vv, int nx, int ny, int border_out);
void orc_bicubic_warp(const double *in, const double *u, const double *v, double *out,
                      int nx, int ny, int border_out);

/* zoom.cpp */
void orc_zoom_size(int nx, int ny, int *nxx, int *nyy, double factor);
int  orc_zoom_out(const double *I, double *Iout, int nx, int ny, double factor);
void orc_zoom_in(const double *I, double *Iout, int nx, int ny, int nxx, int nyy);

/* utils.cpp */
void orc_image_normalization_2(const double *I1, const double *I2, double *I1n, double *I2n, int size);

/* tvl1flow.cpp.  iters (optional, may be NULL) receives the inner-iteration count of every warp
 * (`n` of the reference's verbose line, tvl1flow.cpp:184-188); errs (optional) the final error.
 * max_iter <= 0 selects the reference's MAX_ITERATIONS (300). */
void orc_tvl1_single_scale(const double *I0, const double *I1, double *u1, double *u2, int nx, int ny,
                           double tau, double lambda, double theta, int warps, double epsilon,
                           int verbose, int *iters, double *errs);
int  orc_tvl1_multiscale(const double *I0, const double *I1, double *u1, double *u2, int nx, int ny,
                         double tau, double lambda, double theta, int nscales, double zfactor,
                         int warps, double epsilon, int verbose, int *iters, double *errs);
/* `iters` for the multiscale call is laid out [scale][warp] with scale 0 = finest. */

/* One inner iteration sweep only (for fixed-work CPU timing): runs exactly n_iter iterations of
 * tvl1flow.cpp:113-182 on already-linearised data; returns the last error. */
double orc_tvl1_iterations(double *u1, double *u2, double *p11, double *p12, double *p21, double *p22,
                           const double *I1wx, const double *I1wy, const double *rho_c,
                           const double *grad, int nx, int ny, double tau, double lambda,
                           double theta, int n_iter);

/* checker aids, see ofx_oracle.c: 0 = reference sweep order (default), 1 = the HIP path's colour order,
 * 2 = the HIP path's exact hyperplane-pipelined schedule (bit-identical to 0 by construction) */
void orc_set_sor_order(int order);
/* with order 1: bit s of the mask = pyramid level s is swept in colour order, clear = in the reference's order (default: all set) */
void orc_set_sor_colour_levels(unsigned mask);
/* with order 1: the last `tail` solves (warps / outer x inner iterations) of level 0 keep the reference's order (default 0) */
void orc_set_sor_exact_tail(int tail);
/* Brox, order 3 / order 1 below `levels`: checkerboard of w x h tiles, row-major inside a tile (the HIP path's tolerance mode) */
void orc_set_sor_tile(int w, int h);
void orc_set_sor_wave_levels(int levels);
void orc_set_plane_batch(int sweeps_in_flight);

/* horn_schunck_pyramidal.cpp */
void orc_hs_single_scale(const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                         double alpha, int warps, double TOL, int maxiter, int verbose, int *iters);
int  orc_hs_pyramidal(const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                      double alpha, int nscales, double zfactor, int warps, double TOL, int maxiter,
                      int verbose, int *iters);

/* brox_optic_flow_spatial.cpp + brox_spatial_mask.cpp */
int  orc_brox_spatial(const double *I1, const double *I2, double *u, double *v, int nx, int ny,
                      double alpha, double gamma, int nscales, double nu, double TOL,
                      int inner_iter, int outer_iter, int verbose, int *iters);

/* robust_expo_methods.cpp + robust_expo_smoothness.cpp + robust_expo_generic_tensor.cpp, one channel */
void orc_gaussian_dirichlet(double *I, int nx, int ny, double sigma);
void orc_rexpo_exponential(const double *Ix, const double *Iy, int n, double alpha, double lambda, int method, double *expo);
int  orc_robust_expo(const double *I1, const double *I2, double *u, double *v, int nxx, int nyy, int method, double alpha,
                     double gamma, double lambda, int nscales, double nu, double TOL, int inner_iter, int outer_iter, int verbose,
                     int *iters);

/* brox_optic_flow_temporal.cpp + brox_temporal_mask.cpp (+ operators.cpp centered_gradient3, utils.cpp
 * image_normalization_1) */
void orc_centered_gradient3(const double *in, double *dx, double *dy, double *dz, int nx, int ny, int nz);
void orc_image_normalization_1(const double *I, double *In, int size);
int  orc_brox_temporal(const double *I, double *u, double *v, int nx, int ny, int frames, double alpha, double gamma,
                       int nscales, double nu, double TOL, int inner_iter, int outer_iter, int verbose, int *iters);

/* horn_schunck_classic.cpp */
void orc_hs_classic(double *u, double *v, const double *a, const double *b, int w, int h, int n, double alpha);
double orc_bicubic_at_color(const double *in, double uu, double vv, int nx, int ny, int nz, int k, int border_out);
void orc_getminmax(double *mn, double *mx, const double *x, int n);


/* SURVEY 8(f)4 colour operators: bicubic_interpolation.cpp:381-405, utils.cpp:333-501 */
void orc_bicubic_warp_color(const double *in, const double *u, const double *v, double *out, int nx, int ny, int nz,
                            int border_out);
void orc_image_normalization_2_color(const double *I1, const double *I2, double *I1n, double *I2n, int size, int nz);
void orc_image_normalization_3(double *I0, double *I1, double *I2, int size);
void orc_image_normalization_4(const double *I_1, const double *I0, const double *I1, const double *filtI0, double *I_1n,
                               double *I0n, double *I1n, double *filtI0n, int size);

/* SURVEY 8(f)1 building blocks of TV-L1 with occlusions: utils.cpp:150-213, tvl1occflow_solvers.cpp:56-147,218-337.
 * The chi solver takes its dual variable as explicit state (the reference keeps it in uninitialised statics). */
void orc_median_filtering(double *in, int nx, int ny, int wsize);
void orc_occ_solver_v(const double *u1, const double *u2, double *v1, double *v2, const double *chi, const double *I1wx,
                      const double *I1wy, const double *I_1wx, const double *I_1wy, const double *rho1_c,
                      const double *rho3_c, double *Vfwd_1, double *Vfwd_2, double *Vbck_1, double *Vbck_2,
                      const double *grad1, const double *grad3, double alpha, double theta, double lambda, int nx, int ny);
/* tvl1occflow_tv_rof_box.cpp:22-645 and tvl1occflow_solvers.cpp:150-216 (dual planes as explicit state) */
void orc_rof_box(double *u, const double *f, double *P1, double *P2, const double *g, double lambda, double omega, int nx,
                 int ny, int n_iter);
void orc_occ_solver_u(double *u1, double *u2, const double *v1, const double *v2, const double *chi, const double *g,
                      double theta, double beta, int nx, int ny, double *p11, double *p12, double *p21, double *p22, int n_iter);
void orc_occ_solver_chi(const double *u1, const double *u2, double *chi, const double *I1wx, const double *I1wy,
                        const double *I_1wx, const double *I_1wy, const double *rho1_c, const double *rho3_c,
                        const double *Vfwd_1, const double *Vfwd_2, const double *Vbck_1, const double *Vbck_2,
                        const double *g, double lambda, double theta, double alpha, double beta, double tau_chi,
                        double tau_eta, int nx, int ny, double *eta1, double *eta2, int n_iter);


/* tvl1occflow.cpp:144-481 as the reference computes it on a zero-filled heap (per-level dual state, see ofx_oracle.c) */
void orc_tvl1occ_single_scale(const double *I_1, const double *I0, const double *I1, const double *filtI0, double *u1,
                              double *u2, double *chi, int nx, int ny, double lambda, double alpha, double beta,
                              double theta, int warps, double epsilon, int verbose, double *state, int *iters);
int orc_tvl1occ_multiscale(const double *I_1, const double *I0, const double *I1, const double *filtI0, double *u1, double *u2,
                           double *chi, int nxx, int nyy, double lambda, double alpha, double beta, double theta, int nscales,
                           double zfactor, int warps, double epsilon, int verbose, int *iters);

#ifdef __cplusplus
}
#endif
#endif
