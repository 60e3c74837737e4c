/* ofx_io.c -- see ofx_io.h */
#include "ofx_io.h"

#include <ctype.h>
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

double *ofx_read_image_double(const char *fname, int *w, int *h)
{
    FILE *f = fopen(fname, "rb");
    if (!f) return NULL;
    int c1 = fgetc(f), c2 = fgetc(f), pd = 1;
    float *data = NULL;
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
