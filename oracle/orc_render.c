/* ORACLE (test infrastructure).  Restates the pixel work of the preview / tile renderers -- everything up to, not
 * including, the PNG encoder: cmd/helpers.rs:204-322 (render_rgb_preview, render_rgb_preview_with_stf),
 * infra/render/rgb.rs:7-34 (render_rgb), infra/render/tiles.rs (downsample_2x :41-70, render_tile :72-113,
 * compute_num_levels :137-147, percentile_bounds :149-178, generate_tile_pyramid :180-255, render_tile_rgb(_stf)
 * :257-341, generate_tile_pyramid_rgb_inner :383-481) and infra/ipc.rs:36-148 (raw-f32 buffer with its 16-byte header).
 * See ab_oracle.h for the rules. */
#include "ab_oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline int finite_f(float v) { return isfinite(v); }
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }   /* NaN stays NaN */
static inline uint8_t f32_to_u8_sat(float v) { return !(v > 0.0f) ? 0 : (v >= 255.0f ? 255 : (uint8_t)v); } /* `as u8` */

void orc_preview_dims(size_t rows, size_t cols, size_t max_dim, size_t *ph, size_t *pw) {        /* helpers.rs:283-290 */
    if (rows <= max_dim && cols <= max_dim) { *ph = rows; *pw = cols; return; }
    double scale = (double)max_dim / (double)(rows > cols ? rows : cols);
    *pw = (size_t)fmax(round((double)cols * scale), 1.0);
    *ph = (size_t)fmax(round((double)rows * scale), 1.0);
}

/* one u8 through make_stf_u8_fn (stf.rs:122-145) == apply_stf's per-pixel map */
static uint8_t stf_u8(float v, const orc_stf_params *p, const orc_image_stats *st) {
    uint8_t o;
    orc_apply_stf_u8(&v, 1, p, st, 1, &o);
    return o;
}

/* helpers.rs:204-322 + rgb.rs:7-34: stf == NULL -> (v.clamp(0,1) * 255.0) as u8, else stf[c] / stats[c] per channel.
 * out = ph x pw x 3 interleaved. */
void orc_render_rgb_preview(const float *r, const float *g, const float *b, size_t rows, size_t cols, size_t max_dim,
                            const orc_stf_params *stf, const orc_image_stats *stats, uint8_t *out) {
    size_t ph, pw;
    orc_preview_dims(rows, cols, max_dim, &ph, &pw);
    double y_ratio = 1.0, x_ratio = 1.0;
    if (!(rows <= max_dim && cols <= max_dim)) { y_ratio = (double)rows / (double)ph; x_ratio = (double)cols / (double)pw; }
    const float *ch[3] = {r, g, b};
    for (size_t dy = 0; dy < ph; dy++) {
        size_t sy = (size_t)fmin((double)dy * y_ratio, (double)(rows - 1));
        for (size_t dx = 0; dx < pw; dx++) {
            size_t sx = (size_t)fmin((double)dx * x_ratio, (double)(cols - 1));
            size_t si = sy * cols + sx;
            for (int c = 0; c < 3; c++) {
                float v = ch[c][si];
                out[(dy * pw + dx) * 3 + c] = stf ? stf_u8(v, &stf[c], &stats[c]) : f32_to_u8_sat(clampf(v, 0.0f, 1.0f) * 255.0f);
            }
        }
    }
}

/* ipc.rs:36-148: encode_with_header (max_dim == 0 or both dims <= max_dim) / encode_with_header_downsampled.
 * out must hold 16 + 4 * ph * pw bytes; returns the length. */
size_t orc_ipc_encode_with_header(const float *arr, size_t rows, size_t cols, size_t max_dim, uint8_t *out) {
    size_t ph = rows, pw = cols;
    float mn = FLT_MAX, mx = -FLT_MAX;
    float *px = (float *)(out + 16);
    if (max_dim == 0 || (rows <= max_dim && cols <= max_dim)) {
        for (size_t i = 0; i < rows * cols; i++) {
            float v = arr[i];
            if (finite_f(v)) { if (v < mn) mn = v; if (v > mx) mx = v; }
            px[i] = finite_f(v) ? v : 0.0f;
        }
    } else {
        orc_preview_dims(rows, cols, max_dim, &ph, &pw);
        double y_ratio = (double)rows / (double)ph, x_ratio = (double)cols / (double)pw;
        for (size_t dy = 0; dy < ph; dy++) {
            size_t sy = (size_t)fmin((double)dy * y_ratio, (double)(rows - 1));
            for (size_t dx = 0; dx < pw; dx++) {
                size_t sx = (size_t)fmin((double)dx * x_ratio, (double)(cols - 1));
                float v = arr[sy * cols + sx];
                float clean = finite_f(v) ? v : 0.0f;                  /* min / max see the cleaned value here (:133-135) */
                if (clean < mn) mn = clean;
                if (clean > mx) mx = clean;
                px[dy * pw + dx] = clean;
            }
        }
    }
    float dmin = mn > mx ? 0.0f : mn, dmax = mn > mx ? 1.0f : mx;
    uint32_t w = (uint32_t)pw, h = (uint32_t)ph;
    memcpy(out, &w, 4); memcpy(out + 4, &h, 4); memcpy(out + 8, &dmin, 4); memcpy(out + 12, &dmax, 4);   /* little-endian host */
    return 16 + 4 * ph * pw;
}

size_t orc_tile_compute_num_levels(size_t width, size_t height, size_t tile_size) {              /* tiles.rs:137-147 */
    double max_dim = (double)(width > height ? width : height), ts = (double)tile_size;
    if (max_dim <= ts) return 1;
    size_t levels = (size_t)ceil(log2(max_dim / ts)) + 1;
    return levels > 1 ? levels : 1;
}

void orc_tile_downsample_2x(const float *src, size_t rows, size_t cols, float *out) {            /* tiles.rs:41-70 */
    size_t nr = (rows + 1) / 2, nc = (cols + 1) / 2;
    for (size_t ny = 0; ny < nr; ny++) {
        size_t y0 = ny * 2, y1 = y0 + 1 < rows - 1 ? y0 + 1 : rows - 1;
        for (size_t nx = 0; nx < nc; nx++) {
            size_t x0 = nx * 2, x1 = x0 + 1 < cols - 1 ? x0 + 1 : cols - 1;
            float q[4] = {src[y0 * cols + x0], src[y0 * cols + x1], src[y1 * cols + x0], src[y1 * cols + x1]};
            double sum = 0.0;
            uint32_t count = 0;
            for (int k = 0; k < 4; k++)
                if (finite_f(q[k])) { sum += (double)q[k]; count++; }
            out[ny * nc + nx] = count > 0 ? (float)(sum / (double)count) : 0.0f;
        }
    }
}

void orc_tile_percentile_bounds(const float *slice, size_t n, double low_pct, double high_pct, float *lo, float *hi) {  /* :149-178 */
    float *valid = (float *)malloc((n ? n : 1) * sizeof(float));
    size_t m = 0;
    for (size_t i = 0; i < n; i++)
        if (finite_f(slice[i]) && slice[i] > 1e-7f) valid[m++] = slice[i];
    if (m == 0) {                                             /* find_minmax_simd's portable branch, math/simd.rs:263-271 */
        float mn = FLT_MAX, mx = -FLT_MAX;
        for (size_t i = 0; i < n; i++)
            if (finite_f(slice[i])) { mn = fminf(mn, slice[i]); mx = fmaxf(mx, slice[i]); }
        *lo = mn; *hi = mx;
        free(valid);
        return;
    }
    size_t hi_idx = (size_t)((double)m * high_pct), lo_idx = (size_t)((double)m * low_pct);
    if (hi_idx > m - 1) hi_idx = m - 1;
    if (lo_idx > m - 1) lo_idx = m - 1;
    orc_select_nth_f32(valid, m, hi_idx);
    *hi = valid[hi_idx];
    orc_select_nth_f32(valid, m, lo_idx);
    *lo = valid[lo_idx];
    free(valid);
}

/* render_tile's buffer (tiles.rs:72-113): tile_size^2 bytes, zero outside the image */
void orc_render_tile(const float *src, size_t rows, size_t cols, size_t tx, size_t ty, size_t ts, float gmin, float gmax, uint8_t *buf) {
    memset(buf, 0, ts * ts);
    size_t x0 = tx * ts, y0 = ty * ts;
    size_t x1 = x0 + ts < cols ? x0 + ts : cols, y1 = y0 + ts < rows ? y0 + ts : rows;
    if (x1 <= x0 || y1 <= y0) return;
    float range = fmaxf(gmax - gmin, 1e-10f), inv_range = 255.0f / range;
    for (size_t y = y0; y < y1; y++)
        for (size_t x = x0; x < x1; x++) {
            float v = src[y * cols + x];
            buf[(y - y0) * ts + (x - x0)] = finite_f(v) ? f32_to_u8_sat(clampf(roundf((v - gmin) * inv_range), 0.0f, 255.0f)) : 0;
        }
}

/* render_tile_rgb (:257-298, stf == NULL: (v.clamp(0,1) * 255).round()) / render_tile_rgb_stf (:300-341) */
void orc_render_tile_rgb(const float *r, const float *g, const float *b, size_t rows, size_t cols, size_t tx, size_t ty, size_t ts,
                         const orc_stf_params *stf, const orc_image_stats *stats, uint8_t *buf) {
    memset(buf, 0, ts * ts * 3);
    size_t x0 = tx * ts, y0 = ty * ts;
    size_t x1 = x0 + ts < cols ? x0 + ts : cols, y1 = y0 + ts < rows ? y0 + ts : rows;
    if (x1 <= x0 || y1 <= y0) return;
    const float *ch[3] = {r, g, b};
    for (size_t y = y0; y < y1; y++)
        for (size_t x = x0; x < x1; x++)
            for (int c = 0; c < 3; c++) {
                float v = ch[c][y * cols + x];
                buf[((y - y0) * ts + (x - x0)) * 3 + c] =
                    stf ? stf_u8(v, &stf[c], &stats[c]) : f32_to_u8_sat(roundf(clampf(v, 0.0f, 1.0f) * 255.0f));
            }
}

/* level dims of the pyramid, level 0 = coarsest (tiles.rs:203-246); returns the packed tile bytes for `channels` */
size_t orc_tile_pyramid_layout(size_t rows, size_t cols, size_t ts, size_t channels, orc_tile_level *levels, size_t *num_levels) {
    size_t nl = orc_tile_compute_num_levels(cols, rows, ts), total = 0;
    size_t r = rows, c = cols;
    for (size_t k = 0; k < nl; k++) {                           /* stack index k = max_level - level */
        orc_tile_level *L = &levels[nl - 1 - k];
        L->level = nl - 1 - k; L->width = c; L->height = r;
        L->cols = (c + ts - 1) / ts; L->rows = (r + ts - 1) / ts;
        L->scale_factor = 1.0 / (double)((size_t)1 << k);
        r = (r + 1) / 2; c = (c + 1) / 2;
    }
    for (size_t k = 0; k < nl; k++) {
        levels[k].offset = total;
        total += levels[k].cols * levels[k].rows * ts * ts * channels;
    }
    *num_levels = nl;
    return total;
}

/* generate_tile_pyramid (:180-255) up to the encoder: every tile buffer, level 0 first, tiles in (ty, tx) order */
void orc_generate_tile_pyramid(const float *normalized, size_t rows, size_t cols, size_t ts, uint8_t *tiles, orc_tile_level *levels,
                               size_t *num_levels, float *gmin_out, float *gmax_out) {
    float gmin, gmax;
    orc_tile_percentile_bounds(normalized, rows * cols, 0.001, 0.999, &gmin, &gmax);
    orc_tile_pyramid_layout(rows, cols, ts, 1, levels, num_levels);
    size_t nl = *num_levels;
    const float *cur = normalized;
    float *owned = NULL;
    for (size_t k = 0; k < nl; k++) {
        const orc_tile_level *L = &levels[nl - 1 - k];
        for (size_t ty = 0; ty < L->rows; ty++)
            for (size_t tx = 0; tx < L->cols; tx++)
                orc_render_tile(cur, L->height, L->width, tx, ty, ts, gmin, gmax, tiles + L->offset + (ty * L->cols + tx) * ts * ts);
        if (k + 1 < nl) {
            float *next = (float *)malloc(((L->height + 1) / 2) * ((L->width + 1) / 2) * sizeof(float));
            orc_tile_downsample_2x(cur, L->height, L->width, next);
            free(owned);
            owned = next;
            cur = next;
        }
    }
    free(owned);
    if (gmin_out) *gmin_out = gmin;
    if (gmax_out) *gmax_out = gmax;
}

/* generate_tile_pyramid_rgb_inner (:383-481) up to the encoder */
void orc_generate_tile_pyramid_rgb(const float *r, const float *g, const float *b, size_t rows, size_t cols, size_t ts,
                                   const orc_stf_params *stf, const orc_image_stats *stats, uint8_t *tiles, orc_tile_level *levels,
                                   size_t *num_levels) {
    orc_tile_pyramid_layout(rows, cols, ts, 3, levels, num_levels);
    size_t nl = *num_levels;
    const float *cur[3] = {r, g, b};
    float *owned[3] = {NULL, NULL, NULL};
    for (size_t k = 0; k < nl; k++) {
        const orc_tile_level *L = &levels[nl - 1 - k];
        for (size_t ty = 0; ty < L->rows; ty++)
            for (size_t tx = 0; tx < L->cols; tx++)
                orc_render_tile_rgb(cur[0], cur[1], cur[2], L->height, L->width, tx, ty, ts, stf, stats,
                                    tiles + L->offset + (ty * L->cols + tx) * ts * ts * 3);
        if (k + 1 < nl)
            for (int c = 0; c < 3; c++) {
                float *next = (float *)malloc(((L->height + 1) / 2) * ((L->width + 1) / 2) * sizeof(float));
                orc_tile_downsample_2x(cur[c], L->height, L->width, next);
                free(owned[c]);
                owned[c] = next;
                cur[c] = next;
            }
    }
    for (int c = 0; c < 3; c++) free(owned[c]);
}
