// ofx_sor_tile.h -- tolerance-mode SOR sweeps of Horn-Schunck and Brox (ofx_sor_tile.hip), private to libofx.so.
//
// Both entry points run ONE solve (src/horn_schunck_pyramidal.cpp:143-231, src/brox_optic_flow_spatial.cpp:315-390:
// `while (error > TOL && n < maxiter) sweep`) for the G pairs of a lockstep group and return the reference loop's
// n and error per pair.  They are reached with option sor_exact = 0; the default (exact) mode never comes here.
#pragma once

#include "ofx_internal.h"

// Horn-Schunck, colour order ((i%2, j%2) = (0,0) (0,1) (1,0) (1,1), the order of oracle.set_sor_order(1)), K sweeps per
// launch on LDS tiles.  U0 / U1: the two buffers of the unknowns (every array holds the G pairs back to back); bit g of
// *cur says which one holds pair g's flow, on entry and on return.  K = 0 picks it from the level size.
template <typename T>
int ofx_hs_tile_solve(ofx_ctx *ctx, int G, typename Pix<T>::v2 *U0, typename Pix<T>::v2 *U1, unsigned *cur,
                      const typename Pix<T>::v2 *A, const T *Dif, int nx, int ny, double alpha2, double TOL, int maxiter, int K,
                      int *niter, double *error, float *ms);

// Brox, checkerboard of 64-row tiles with the reference's order inside a tile (ofx_sor_tile.hip, k_brox_wave), on BAND-SKEWED
// planes: ofx_band_plane_elems(nx, ny) elements per pair and plane, converted from / to row-major by ofx_band_copy<V, IN>
// (IN = true: row-major -> band).  DUb is updated in place.
size_t ofx_band_plane_elems(int nx, int ny);
template <typename V, bool IN> int ofx_band_copy(ofx_ctx *ctx, const V *src, V *dst, int nx, int ny, int G);
template <typename T>
int ofx_brox_wave_solve(ofx_ctx *ctx, int G, typename Pix<T>::v2 *DUb, const typename Pix<T>::v4 *COb, const T *Dmb, const T *Psb,
                        int nx, int ny, double alpha, double TOL, int maxiter, int *niter, double *error, float *ms);

// Brox, the levels below those: red-black sweeps ((i + j) even first, the order of k_brox_sor and oracle.set_sor_order(1)), K per
// launch on LDS tiles (k_brox_tile).  DU0 holds (du, dv) on entry and on return; DU1 is the second buffer of the ping-pong.
template <typename T>
int ofx_brox_tile_solve(ofx_ctx *ctx, int G, typename Pix<T>::v2 *DU0, typename Pix<T>::v2 *DU1, const typename Pix<T>::v4 *CO,
                        const T *Dm, const T *Ps, int nx, int ny, double alpha, double TOL, int maxiter, int K, int *niter, double *error,
                        float *ms);
