/* ORACLE (test infrastructure).  Restates the pixel codecs of infra/fits: decode_pixels (reader.rs:42-101),
 * is_identity_scaling (:36-39), write_f32 / i16 / f64_slice_as_be (writer.rs:82-135) and compute_bzero_bscale
 * (:143-159).  Header parsing, mmap and file IO are outside the hot path.  See ab_oracle.h for the rules. */
#include "ab_oracle.h"
#include <math.h>
#include <string.h>

static int is_identity_scaling(double bscale, double bzero) { return fabs(bscale - 1.0) < 1e-15 && fabs(bzero) < 1e-15; }

/* returns the number of pixels decoded (0 for an unknown BITPIX, as the reference's empty Vec) */
size_t orc_fits_decode_pixels(const uint8_t *data, size_t nbytes, int64_t bitpix, double bscale, double bzero, float *out) {
    const int identity = is_identity_scaling(bscale, bzero);
    size_t n = 0;
    switch (bitpix) {
    case 8:
        n = nbytes;
        for (size_t i = 0; i < n; i++) out[i] = identity ? (float)data[i] : (float)((double)data[i] * bscale + bzero);
        break;
    case 16:
        n = nbytes / 2;
        for (size_t i = 0; i < n; i++) {
            int16_t v = (int16_t)(((uint16_t)data[2 * i] << 8) | data[2 * i + 1]);
            out[i] = identity ? (float)v : (float)((double)v * bscale + bzero);
        }
        break;
    case 32:
        n = nbytes / 4;
        for (size_t i = 0; i < n; i++) {
            int32_t v = (int32_t)(((uint32_t)data[4 * i] << 24) | ((uint32_t)data[4 * i + 1] << 16) | ((uint32_t)data[4 * i + 2] << 8) | data[4 * i + 3]);
            out[i] = identity ? (float)v : (float)((double)v * bscale + bzero);
        }
        break;
    case -32:
        n = nbytes / 4;
        for (size_t i = 0; i < n; i++) {
            uint32_t u = ((uint32_t)data[4 * i] << 24) | ((uint32_t)data[4 * i + 1] << 16) | ((uint32_t)data[4 * i + 2] << 8) | data[4 * i + 3];
            float v;
            memcpy(&v, &u, 4);
            out[i] = identity ? v : (float)((double)v * bscale + bzero);
        }
        break;
    case -64:
        n = nbytes / 8;
        for (size_t i = 0; i < n; i++) {
            uint64_t u = 0;
            for (int k = 0; k < 8; k++) u = (u << 8) | data[8 * i + k];
            double v;
            memcpy(&v, &u, 8);
            out[i] = identity ? (float)v : (float)(v * bscale + bzero);
        }
        break;
    default:
        n = 0;
    }
    return n;
}

/* writer.rs:143-159 */
void orc_fits_compute_bzero_bscale(const float *data, size_t n, double *bzero, double *bscale) {
    double dmin = INFINITY, dmax = -INFINITY;
    for (size_t i = 0; i < n; i++) {
        double v = (double)data[i];
        if (isfinite(v)) {
            if (v < dmin) dmin = v;
            if (v > dmax) dmax = v;
        }
    }
    if (!isfinite(dmin) || !isfinite(dmax) || fabs(dmax - dmin) < 1e-30) { *bzero = 32768.0; *bscale = 1.0; return; }
    *bscale = (dmax - dmin) / 65535.0;
    *bzero = dmin + *bscale * 32768.0;
}

/* writer.rs:82-135: big-endian f32 (bitpix -32), i16 with BZERO/BSCALE (16), f64 (-64).  Returns bytes written, 0 if
 * the BITPIX is not one the writer supports. */
size_t orc_fits_encode_pixels(const float *data, size_t n, int32_t bitpix, double bzero, double bscale, uint8_t *out) {
    if (bitpix == -32) {
        for (size_t i = 0; i < n; i++) {
            uint32_t u;
            memcpy(&u, &data[i], 4);
            out[4 * i] = (uint8_t)(u >> 24); out[4 * i + 1] = (uint8_t)(u >> 16); out[4 * i + 2] = (uint8_t)(u >> 8); out[4 * i + 3] = (uint8_t)u;
        }
        return 4 * n;
    }
    if (bitpix == 16) {
        for (size_t i = 0; i < n; i++) {
            double physical = ((double)data[i] - bzero) / bscale;
            double c = physical < -32768.0 ? -32768.0 : (physical > 32767.0 ? 32767.0 : physical);   /* f64::clamp: NaN stays NaN */
            double r = round(c);
            int16_t v = (r != r) ? 0 : (int16_t)r;                                                   /* NaN as i16 = 0 */
            out[2 * i] = (uint8_t)((uint16_t)v >> 8);
            out[2 * i + 1] = (uint8_t)((uint16_t)v & 0xff);
        }
        return 2 * n;
    }
    if (bitpix == -64) {
        for (size_t i = 0; i < n; i++) {
            double d = (double)data[i];
            uint64_t u;
            memcpy(&u, &d, 8);
            for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(u >> (56 - 8 * k));
        }
        return 8 * n;
    }
    return 0;
}
