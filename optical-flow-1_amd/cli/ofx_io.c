/* ofx_io.c -- see ofx_io.h */
#include "ofx_io.h"

#include <ctype.h>
#include <dlfcn.h>
#include <math.h>
#include <setjmp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int ofx_has_suffix(const char *s, const char *suffix)
{
    const size_t n = strlen(s), m = strlen(suffix);
    return n >= m && strcmp(s + n - m, suffix) == 0;
}

static void skip_space_and_comments(FILE *f)
{
    int c;
    for (;;) {
        do { c = fgetc(f); } while (c != EOF && isspace(c));
        if (c == '#') {
            do { c = fgetc(f); } while (c != EOF && c != '\n');
            continue;
        }
        if (c != EOF) ungetc(c, f);
        return;
    }
}

static int read_header_int(FILE *f, int *v)
{
    skip_space_and_comments(f);
    return fscanf(f, "%d", v) == 1;
}

/* PGM / PPM body -> float samples */
static float *read_pnm(FILE *f, int kind, int *w, int *h, int *pd)
{
    int m;
    if (!read_header_int(f, w) || !read_header_int(f, h) || !read_header_int(f, &m)) return NULL;
    if (!isspace(fgetc(f))) return NULL;                 /* exactly one whitespace before the raster */
    if (*w <= 0 || *h <= 0 || m <= 0 || m >= 0x10000) return NULL;
    *pd = (kind == 3 || kind == 6) ? 3 : 1;
    const size_t nn = (size_t) *w * *h * *pd;
    float *data = (float *) malloc(nn * sizeof(float));
    if (!data) return NULL;
    const int ascii = (kind == 2 || kind == 3);
    for (size_t i = 0; i < nn; i++) {
        if (ascii) {
            float c;
            if (fscanf(f, "%f ", &c) != 1) { free(data); return NULL; }
            data[i] = c;
        } else if (m < 0x100) {
            const int c = fgetc(f);
            if (c == EOF) { free(data); return NULL; }
            data[i] = (float) c;
        } else {
            const int hi = fgetc(f), lo = fgetc(f);      /* big-endian shorts, iio.cpp:1744-1755 */
            if (hi == EOF || lo == EOF) { free(data); return NULL; }
            data[i] = (float) hi * 256 + (float) lo;
        }
    }
    return data;
}

static float *read_pfm(FILE *f, int colour, int *w, int *h, int *pd)
{
    float scale;
    if (!isspace(fgetc(f))) return NULL;
    if (fscanf(f, "%d %d\n%g", w, h, &scale) != 3) return NULL;
    if (!isspace(fgetc(f))) return NULL;
    if (*w <= 0 || *h <= 0) return NULL;
    *pd = colour ? 3 : 1;
    const size_t nn = (size_t) *w * *h * *pd;
    float *data = (float *) malloc(nn * sizeof(float));
    if (!data) return NULL;
    if (fread(data, sizeof(float), nn, f) != nn) { free(data); return NULL; }
    return data;                                          /* no flip, no byte swap (iio.cpp:2216) */
}

/* ---- PNG through the system's libpng16, bound at run time ---------------------------------------------
 * The reference decodes PNG with libpng (iio.cpp:1365-1438): png_read_png with PNG_TRANSFORM_PACKING |
 * PNG_TRANSFORM_EXPAND, i.e. 1/2/4-bit gray scaled to 8 bit, palettes expanded to RGB, tRNS to an alpha channel;
 * 8-bit samples become IIO_TYPE_CHAR (= uint8, :493-494), 16-bit samples host-order uint16 (:1421-1432).
 * iio_read_image_double (:3580-3605) then collapses exactly-3-channel images to gray
 * (uint8)(.299 R + .587 G + .114 B) -- double arithmetic truncated to uint8 (:1100-1108; uint16 colour is
 * "uncolorize type not supported", :1119-1121) -- and rejects every other channel count but 1 ("non-scalar image").
 * There are no libpng headers in the build image, so the handful of entry points used is declared here and the
 * library is dlopen'ed; without it PNG input fails with a message, the other formats are unaffected. */
typedef struct ofx_png_struct ofx_png_struct;
typedef struct ofx_png_info ofx_png_info;
#define OFX_PNG_TRANSFORM_PACKING 0x0004
#define OFX_PNG_TRANSFORM_EXPAND 0x0010
struct ofx_png_api {
    void *lib;
    const char *(*get_libpng_ver)(const ofx_png_struct *);
    ofx_png_struct *(*create_read_struct)(const char *, void *, void *, void *);
    ofx_png_info *(*create_info_struct)(ofx_png_struct *);
    jmp_buf *(*set_longjmp_fn)(ofx_png_struct *, void (*)(jmp_buf, int), size_t);
    void (*init_io)(ofx_png_struct *, FILE *);
    void (*read_png)(ofx_png_struct *, ofx_png_info *, int, void *);
    uint32_t (*get_image_width)(const ofx_png_struct *, const ofx_png_info *);
    uint32_t (*get_image_height)(const ofx_png_struct *, const ofx_png_info *);
    unsigned char (*get_channels)(const ofx_png_struct *, const ofx_png_info *);
    unsigned char (*get_bit_depth)(const ofx_png_struct *, const ofx_png_info *);
    unsigned char **(*get_rows)(const ofx_png_struct *, const ofx_png_info *);
    void (*destroy_read_struct)(ofx_png_struct **, ofx_png_info **, ofx_png_info **);
    /* writer */
    ofx_png_struct *(*create_write_struct)(const char *, void *, void *, void *);
    void (*set_IHDR)(const ofx_png_struct *, ofx_png_info *, uint32_t, uint32_t, int, int, int, int, int);
    void (*set_rows)(const ofx_png_struct *, ofx_png_info *, unsigned char **);
    void (*write_png)(ofx_png_struct *, ofx_png_info *, int, void *);
    void (*destroy_write_struct)(ofx_png_struct **, ofx_png_info **);
};

static const struct ofx_png_api *png_api(void)
{
    static struct ofx_png_api api;
    static int state = 0;                                  /* 0 untried, 1 ok, -1 unavailable */
    if (state) return state > 0 ? &api : NULL;
    state = -1;
    api.lib = dlopen("libpng16.so.16", RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) api.lib = dlopen("libpng16.so", RTLD_NOW | RTLD_LOCAL);
    if (!api.lib) return NULL;
#define OFX_PNG_SYM(field, name)                                                                  \
    do {                                                                                          \
        *(void **) (&api.field) = dlsym(api.lib, name);                                           \
        if (!api.field) return NULL;                                                              \
    } while (0)
    OFX_PNG_SYM(get_libpng_ver, "png_get_libpng_ver");
    OFX_PNG_SYM(create_read_struct, "png_create_read_struct");
    OFX_PNG_SYM(create_info_struct, "png_create_info_struct");
    OFX_PNG_SYM(set_longjmp_fn, "png_set_longjmp_fn");
    OFX_PNG_SYM(init_io, "png_init_io");
    OFX_PNG_SYM(read_png, "png_read_png");
    OFX_PNG_SYM(get_image_width, "png_get_image_width");
    OFX_PNG_SYM(get_image_height, "png_get_image_height");
    OFX_PNG_SYM(get_channels, "png_get_channels");
    OFX_PNG_SYM(get_bit_depth, "png_get_bit_depth");
    OFX_PNG_SYM(get_rows, "png_get_rows");
    OFX_PNG_SYM(destroy_read_struct, "png_destroy_read_struct");
    OFX_PNG_SYM(create_write_struct, "png_create_write_struct");
    OFX_PNG_SYM(set_IHDR, "png_set_IHDR");
    OFX_PNG_SYM(set_rows, "png_set_rows");
    OFX_PNG_SYM(write_png, "png_write_png");
    OFX_PNG_SYM(destroy_write_struct, "png_destroy_write_struct");
#undef OFX_PNG_SYM
    state = 1;
    return &api;
}

/* f is positioned at the start of the file.  Returns gray samples as double, or NULL. */
static double *read_png(FILE *f, int *w, int *h)
{
    const struct ofx_png_api *P = png_api();
    if (!P) {
        fprintf(stderr, "PNG input needs libpng16.so.16 at run time (not found)\n");
        return NULL;
    }
    ofx_png_struct *pp = P->create_read_struct(P->get_libpng_ver(NULL), NULL, NULL, NULL);
    if (!pp) return NULL;
    ofx_png_info *pi = P->create_info_struct(pp);
    double *volatile out = NULL;
    if (!pi) { P->destroy_read_struct(&pp, NULL, NULL); return NULL; }
    jmp_buf *jb = P->set_longjmp_fn(pp, longjmp, sizeof(jmp_buf));
    if (!jb || setjmp(*jb)) {                              /* libpng reports a broken file by longjmp */
        P->destroy_read_struct(&pp, &pi, NULL);
        free(out);
        return NULL;
    }
    P->init_io(pp, f);
    P->read_png(pp, pi, OFX_PNG_TRANSFORM_PACKING | OFX_PNG_TRANSFORM_EXPAND, NULL);      /* iio.cpp:1386-1390 */
    const uint32_t ww = P->get_image_width(pp, pi), hh = P->get_image_height(pp, pi);
    const int ch = P->get_channels(pp, pi), depth = P->get_bit_depth(pp, pi);
    unsigned char **rows = P->get_rows(pp, pi);
    const int ok = rows && ww > 0 && hh > 0 && ((depth == 8 && (ch == 1 || ch == 3)) || (depth == 16 && ch == 1));
    if (!ok) fprintf(stderr, "PNG with %d channel(s) of %d bits: not a scalar image for the reference either\n", ch, depth);
    if (ok) out = (double *) malloc((size_t) ww * hh * sizeof(double));
    if (ok && out) {
        for (uint32_t j = 0; j < hh; j++)
            for (uint32_t i = 0; i < ww; i++) {
                const unsigned char *b = rows[j] + (size_t) i * ch * (depth / 8);
                double v;
                if (depth == 16) v = (double) (uint16_t) ((b[0] << 8) | b[1]);                  /* :1425-1429 */
                else if (ch == 1) v = (double) b[0];
                else v = (double) (uint8_t) (.299 * b[0] + .587 * b[1] + .114 * b[2]);          /* :1104-1105 */
                out[(size_t) j * ww + i] = v;
            }
        *w = (int) ww;
        *h = (int) hh;
    }
    P->destroy_read_struct(&pp, &pi, NULL);
    return out;
}

double *ofx_read_image_double(const char *fname, int *w, int *h)
{
    FILE *f = fopen(fname, "rb");
    if (!f) return NULL;
    int c1 = fgetc(f), c2 = fgetc(f), pd = 1;
    float *data = NULL;
    if (c1 == 0x89 && c2 == 'P') {                          /* PNG signature 89 50 4E 47 ... */
        rewind(f);
        double *png = read_png(f, w, h);
        fclose(f);
        return png;
    }
    if (c1 == 'P' && c2 >= '2' && c2 <= '6' && c2 != '4') data = read_pnm(f, c2 - '0', w, h, &pd);
    else if (c1 == 'P' && (c2 == 'f' || c2 == 'F')) data = read_pfm(f, c2 == 'F', w, h, &pd);
    fclose(f);
    if (!data) return NULL;
    const size_t n = (size_t) *w * *h;
    double *out = (double *) malloc(n * sizeof(double));
    if (!out) { free(data); return NULL; }
    for (size_t i = 0; i < n; i++) {
        if (pd == 3) {
            const float g = .299 * data[3 * i] + .587 * data[3 * i + 1] + .114 * data[3 * i + 2];
            out[i] = g;
        } else {
            out[i] = data[i];
        }
    }
    free(data);
    return out;
}

/* 8-bit gray PNG with libpng's defaults, as iio_save_image_as_png writes it (iio.cpp:2539-2606: png_set_IHDR(8, GRAY, no
 * interlace), png_set_rows, png_write_png(PNG_TRANSFORM_IDENTITY)) -- the same bytes when both sides use the same libpng */
static int write_png_gray8(const char *fname, const uint8_t *data, int w, int h)
{
    const struct ofx_png_api *P = png_api();
    if (!P) {
        fprintf(stderr, "PNG output needs libpng16.so.16 at run time (not found)\n");
        return 1;
    }
    FILE *f = fopen(fname, "wb");
    if (!f) return 1;
    ofx_png_struct *pp = P->create_write_struct(P->get_libpng_ver(NULL), NULL, NULL, NULL);
    ofx_png_info *pi = pp ? P->create_info_struct(pp) : NULL;
    unsigned char **volatile rows = NULL;
    int rc = 1;
    if (pp && pi) {
        jmp_buf *jb = P->set_longjmp_fn(pp, longjmp, sizeof(jmp_buf));
        if (jb && !setjmp(*jb)) {
            rows = (unsigned char **) malloc((size_t) h * sizeof(*rows));
            if (rows) {
                for (int i = 0; i < h; i++) rows[i] = (unsigned char *) data + (size_t) i * w;
                P->init_io(pp, f);
                P->set_IHDR(pp, pi, (uint32_t) w, (uint32_t) h, 8, 0 /* GRAY */, 0 /* no interlace */, 0, 0);
                P->set_rows(pp, pi, rows);
                P->write_png(pp, pi, 0 /* PNG_TRANSFORM_IDENTITY */, NULL);
                rc = 0;
            }
        }
    }
    if (pp) P->destroy_write_struct(&pp, pi ? &pi : NULL);
    free(rows);
    if (fclose(f)) rc = 1;
    return rc;
}

/* iio_save_image_float for a one-channel image whose samples are all integers in [0, 255] ("these floats are actually
 * bytes", iio.cpp:3620-3636,3698-3710: converted to uint8 first): .png / .PNG -> 8-bit gray PNG (:3752-3773), any other
 * name -> PGM, ASCII P2 up to 10000 pixels and binary P5 above (:3838-3853).  TIFF names and images with other sample
 * values (which the reference would send through its sample conversion or libtiff) are refused: returns 2. */
int ofx_write_gray_bytes(const char *fname, const float *x, int w, int h)
{
    const size_t n = (size_t) w * h;
    if (ofx_has_suffix(fname, ".tiff") || ofx_has_suffix(fname, ".tif") || ofx_has_suffix(fname, ".TIFF") ||
        ofx_has_suffix(fname, ".TIF") || !strncmp(fname, "TIFF:", 5))
        return 2;
    for (size_t i = 0; i < n; i++)
        if (!(x[i] == floorf(x[i]) && x[i] >= 0 && x[i] < 256)) return 2;
    uint8_t *b = (uint8_t *) malloc(n ? n : 1);
    if (!b) return 1;
    for (size_t i = 0; i < n; i++) b[i] = (uint8_t) x[i];
    int rc;
    if (!strncmp(fname, "PNG:", 4)) rc = write_png_gray8(fname + 4, b, w, h);
    else if (ofx_has_suffix(fname, ".png") || ofx_has_suffix(fname, ".PNG")) rc = write_png_gray8(fname, b, w, h);
    else {
        FILE *f = fopen(fname, "w");
        rc = f ? 0 : 1;
        if (f && n <= 10000) {
            fprintf(f, "P2\n%d %d\n255\n", w, h);
            for (size_t i = 0; i < n; i++) fprintf(f, "%d\n", (int) b[i]);
        } else if (f) {
            fprintf(f, "P5\n%d %d\n255\n", w, h);
            if (fwrite(b, n, 1, f) != 1) rc = 1;
        }
        if (f && fclose(f)) rc = 1;
    }
    free(b);
    return rc;
}

int ofx_write_flo(const char *fname, const float *uv, int w, int h)
{
    FILE *f = fopen(fname, "wb");
    if (!f) return 1;
    const char magic[4] = {'P', 'I', 'E', 'H'};
    const uint32_t ww = (uint32_t) w, hh = (uint32_t) h;
    int ok = fwrite(magic, 4, 1, f) == 1 && fwrite(&ww, 4, 1, f) == 1 && fwrite(&hh, 4, 1, f) == 1;
    ok = ok && fwrite(uv, sizeof(float), (size_t) w * h * 2, f) == (size_t) w * h * 2;
    return (fclose(f) == 0 && ok) ? 0 : 1;
}

float *ofx_read_flo(const char *fname, int *w, int *h)
{
    FILE *f = fopen(fname, "rb");
    if (!f) return NULL;
    char magic[4];
    uint32_t ww, hh;
    float *uv = NULL;
    if (fread(magic, 4, 1, f) == 1 && memcmp(magic, "PIEH", 4) == 0 && fread(&ww, 4, 1, f) == 1 &&
        fread(&hh, 4, 1, f) == 1 && ww > 0 && hh > 0) {
        const size_t n = (size_t) ww * hh * 2;
        uv = (float *) malloc(n * sizeof(float));
        if (uv && fread(uv, sizeof(float), n, f) != n) { free(uv); uv = NULL; }
        *w = (int) ww;
        *h = (int) hh;
    }
    fclose(f);
    return uv;
}
