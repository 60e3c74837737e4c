/* ofx_io.h -- dependency-free image input and .flo output for the command-line front-ends.
 *
 * Observable behaviour follows the reference's iio layer for the formats that matter here
 * (the reference's iio.cpp itself needs libpng/libjpeg/libtiff and is out of scope):
 *   PGM/PPM (P2 P5 P3 P6): samples are read as floats, maxval only selects 1- or 2-byte samples
 *                          (src/iio.cpp:1775-1795, :1808-1881); '#' comments allowed in the header.
 *   PFM (Pf / PF)        : raw floats, NO vertical flip and NO endianness swap (src/iio.cpp:2194-2229).
 *   3-channel input      : collapsed to gray .299 R + .587 G + .114 B, rounded to float
 *                          (src/iio.cpp:1110-1118), then widened to double (src/iio.cpp:3579-3605).
 *   PNG                  : through the system's libpng16 bound at run time (see ofx_io.c); gray truncation as iio.cpp:1100-1108.
 *   .flo output          : "PIEH", uint32 width, uint32 height, w*h interleaved (u,v) float32,
 *                          host byte order (src/iio.cpp:2753-2777).
 */
#ifndef OFX_IO_H
#define OFX_IO_H

#ifdef __cplusplus
extern "C" {
#endif

/* returns a malloc'ed w*h double image (caller frees) or NULL */
double *ofx_read_image_double(const char *fname, int *w, int *h);
/* uv = w*h interleaved (u,v) pairs; returns 0 on success */
int ofx_write_flo(const char *fname, const float *uv, int w, int h);
/* one-channel float image whose samples are all bytes (iio_save_image_float, src/iio.cpp:3698-3710): .png -> 8-bit gray PNG
 * through libpng16 (dlopen'ed), other names -> PGM (P2 up to 10000 pixels, P5 above).  0 ok, 1 I/O error, 2 not
 * representable here (TIFF name, non-byte samples) */
int ofx_write_gray_bytes(const char *fname, const float *x, int w, int h);
/* returns a malloc'ed w*h*2 float array or NULL */
float *ofx_read_flo(const char *fname, int *w, int *h);
/* 1 if `s` ends with `suffix` */
int ofx_has_suffix(const char *s, const char *suffix);

#ifdef __cplusplus
}
#endif
#endif
