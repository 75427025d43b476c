/* ORACLE (test infrastructure).  Restates core/alignment/affine.rs: align_channel_affine
 * (:129-212), check_transform_sanity (:214-241), fallback_phase_correlation (:243-270),
 * top_n_stars (:272-277), build_triangles (:279-318), match_triangles (:320-384),
 * sort_triangle_vertices (:386-398), ransac_affine (:400-517), fit_affine / solve_3x3_ls /
 * solve_3x3 (:519-595), fit_rigid (:597-642), compute_residual (:644-656), dist (:658-661).
 * See ab_oracle.h for the rules.
 *
 * PINNED NONDETERMINISM of the reference (hard part 3):
 *  - vote pairs come out of a std HashMap and are stable-sorted by votes only (:351-360): ties are
 *    in random order per process.  Pinned: votes descending, then (ref index, tgt index) ascending.
 *  - RANSAC splits 2000 iterations over rayon::current_num_threads() workers with per-worker
 *    xorshift seeds (:410-416), so the reference's answer depends on the host's core count.
 *    Pinned: num_threads is an explicit argument. */
#include "ab_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAX_STARS 120
#define TRIANGLE_TOLERANCE 0.02
#define MIN_MATCHES_AFFINE 6
#define MIN_MATCHES_RIGID 4
#define RANSAC_ITERATIONS 2000
#define RANSAC_INLIER_PX 3.0
#define DETECTION_SIGMA 3.5
#define MIN_TRIANGLE_SIDE 15.0
#define MIN_VOTES 1u
#define MIN_INLIER_RATIO 0.20
#define MAX_RESIDUAL_PX 5.0
#define MAX_OFFSET_FRACTION 0.40
#define MAX_ROTATION_DEG 30.0
#define MIN_SCALE 0.70
#define MAX_SCALE 1.40

enum { M_AFFINE = 0, M_RIGID = 1, M_PC = 2, M_IDENTITY = 3 };

typedef struct { size_t idx[3]; double ratio_mid, ratio_long; } tri_t;

static double dist2(const double *xy, size_t a, size_t b) {        /* :658-661 */
    double dx = xy[2 * a] - xy[2 * b], dy = xy[2 * a + 1] - xy[2 * b + 1];
    return sqrt(dx * dx + dy * dy);
}

static void sort3(double *s) {                                        /* stable for 3 elements */
    if (s[1] < s[0]) { double t = s[0]; s[0] = s[1]; s[1] = t; }
    if (s[2] < s[1]) { double t = s[1]; s[1] = s[2]; s[2] = t; if (s[1] < s[0]) { t = s[0]; s[0] = s[1]; s[1] = t; } }
}

/* :279-318 */
static tri_t *build_triangles(const double *xy, size_t n, size_t *count) {
    *count = 0;
    if (n < 3) return NULL;
    size_t limit = n < 60 ? n : 60;
    tri_t *tris = (tri_t *)malloc(limit * limit * limit / 6 * sizeof(tri_t) + sizeof(tri_t));
    size_t m = 0;
    for (size_t i = 0; i < limit; i++)
        for (size_t j = i + 1; j < limit; j++)
            for (size_t k = j + 1; k < limit; k++) {
                double s[3] = {dist2(xy, i, j), dist2(xy, j, k), dist2(xy, i, k)};
                sort3(s);
                if (s[0] < MIN_TRIANGLE_SIDE) continue;
                tris[m].idx[0] = i; tris[m].idx[1] = j; tris[m].idx[2] = k;
                tris[m].ratio_mid = s[1] / s[0];
                tris[m].ratio_long = s[2] / s[0];
                m++;
            }
    *count = m;
    return tris;
}

/* :386-398: vertex ids ordered by the length of the opposite side (stable) */
static void sort_triangle_vertices(const double *xy, const size_t idx[3], size_t out[3]) {
    size_t v[3] = {idx[0], idx[1], idx[2]};
    double d[3] = {dist2(xy, idx[1], idx[2]), dist2(xy, idx[0], idx[2]), dist2(xy, idx[0], idx[1])};
    for (int a = 1; a < 3; a++)                                        /* insertion sort = stable */
        for (int b = a; b > 0 && d[b] < d[b - 1]; b--) {
            double td = d[b]; d[b] = d[b - 1]; d[b - 1] = td;
            size_t tv = v[b]; v[b] = v[b - 1]; v[b - 1] = tv;
        }
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
}

typedef struct { size_t ri, ti; uint32_t votes; } pair_t;
static int cmp_pair(const void *pa, const void *pb) {
    const pair_t *a = (const pair_t *)pa, *b = (const pair_t *)pb;
    if (a->votes != b->votes) return a->votes > b->votes ? -1 : 1;
    if (a->ri != b->ri) return a->ri < b->ri ? -1 : 1;
    return a->ti < b->ti ? -1 : (a->ti > b->ti ? 1 : 0);
}

/* :320-384; matches out: rows of (rx, ry, tx, ty) */
static size_t match_triangles(const double *rxy, size_t nr, const double *txy, size_t nt, const tri_t *rt, size_t nrt,
                              const tri_t *tt, size_t ntt, double *matches) {
    uint32_t *votes = (uint32_t *)calloc(nr * nt, sizeof(uint32_t));
    for (size_t a = 0; a < nrt; a++)
        for (size_t b = 0; b < ntt; b++) {
            double d_mid = fabs(rt[a].ratio_mid - tt[b].ratio_mid), d_long = fabs(rt[a].ratio_long - tt[b].ratio_long);
            if (d_mid > TRIANGLE_TOLERANCE || d_long > TRIANGLE_TOLERANCE) continue;
            size_t rs[3], ts[3];
            sort_triangle_vertices(rxy, rt[a].idx, rs);
            sort_triangle_vertices(txy, tt[b].idx, ts);
            for (int p = 0; p < 3; p++) votes[rs[p] * nt + ts[p]] += 1;
        }
    size_t np = 0;
    pair_t *pairs = (pair_t *)malloc((nr * nt ? nr * nt : 1) * sizeof(pair_t));
    for (size_t r = 0; r < nr; r++)
        for (size_t t = 0; t < nt; t++)
            if (votes[r * nt + t]) { pairs[np].ri = r; pairs[np].ti = t; pairs[np].votes = votes[r * nt + t]; np++; }
    qsort(pairs, np, sizeof(pair_t), cmp_pair);
    unsigned char *used_r = (unsigned char *)calloc(nr ? nr : 1, 1), *used_t = (unsigned char *)calloc(nt ? nt : 1, 1);
    size_t nm = 0;
    for (size_t i = 0; i < np; i++) {
        if (pairs[i].votes < MIN_VOTES) break;
        if (used_r[pairs[i].ri] || used_t[pairs[i].ti]) continue;
        used_r[pairs[i].ri] = 1; used_t[pairs[i].ti] = 1;
        matches[4 * nm] = rxy[2 * pairs[i].ri]; matches[4 * nm + 1] = rxy[2 * pairs[i].ri + 1];
        matches[4 * nm + 2] = txy[2 * pairs[i].ti]; matches[4 * nm + 3] = txy[2 * pairs[i].ti + 1];
        nm++;
    }
    free(votes); free(pairs); free(used_r); free(used_t);
    return nm;
}

/* :556-595 */
static int solve_3x3(double a[3][3], const double b[3], double x[3]) {
    double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                 a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    if (fabs(det) < 1e-12) return 0;
    double inv_det = 1.0 / det;
    double inv[3][3] = {
        {(a[1][1] * a[2][2] - a[1][2] * a[2][1]) * inv_det, (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * inv_det,
         (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * inv_det},
        {(a[1][2] * a[2][0] - a[1][0] * a[2][2]) * inv_det, (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * inv_det,
         (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * inv_det},
        {(a[1][0] * a[2][1] - a[1][1] * a[2][0]) * inv_det, (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * inv_det,
         (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * inv_det}};
    for (int i = 0; i < 3; i++) x[i] = inv[i][0] * b[0] + inv[i][1] * b[1] + inv[i][2] * b[2];
    return 1;
}

/* :538-554 */
static int solve_3x3_ls(const double *m, size_t n, int solve_x, double out[3]) {
    double ata[3][3] = {{0}}, atb[3] = {0};
    for (size_t k = 0; k < n; k++) {
        double target = solve_x ? m[4 * k + 2] : m[4 * k + 3];
        double row[3] = {m[4 * k], m[4 * k + 1], 1.0};
        for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) ata[i][j] += row[i] * row[j];
            atb[i] += row[i] * target;
        }
    }
    return solve_3x3(ata, atb, out);
}

/* :519-536 */
int orc_fit_affine(const double *m, size_t n, double t[6]) {
    if (n < 3) return 0;
    double ab[3], cd[3];
    if (!solve_3x3_ls(m, n, 1, ab)) return 0;
    if (!solve_3x3_ls(m, n, 0, cd)) return 0;
    t[0] = ab[0]; t[1] = ab[1]; t[2] = ab[2]; t[3] = cd[0]; t[4] = cd[1]; t[5] = cd[2];
    return 1;
}

/* :597-642 */
int orc_fit_rigid(const double *m, size_t n, double t[6]) {
    if (n < 2) return 0;
    double rcx = 0.0, rcy = 0.0, tcx = 0.0, tcy = 0.0;
    for (size_t k = 0; k < n; k++) { rcx += m[4 * k]; rcy += m[4 * k + 1]; tcx += m[4 * k + 2]; tcy += m[4 * k + 3]; }
    double nf = (double)n;
    rcx /= nf; rcy /= nf; tcx /= nf; tcy /= nf;
    double num = 0.0, den = 0.0;
    for (size_t k = 0; k < n; k++) {
        double drx = m[4 * k] - rcx, dry = m[4 * k + 1] - rcy, dtx = m[4 * k + 2] - tcx, dty = m[4 * k + 3] - tcy;
        num += drx * dty - dry * dtx;
        den += drx * dtx + dry * dty;
    }
    double theta = atan2(num, den), cos_t = cos(theta), sin_t = sin(theta);
    t[0] = cos_t; t[1] = -sin_t; t[2] = tcx - cos_t * rcx + sin_t * rcy;
    t[3] = sin_t; t[4] = cos_t; t[5] = tcy - sin_t * rcx - cos_t * rcy;
    return 1;
}

static double point_err(const double t[6], const double *m) {
    double px = t[0] * m[0] + t[1] * m[1] + t[2], py = t[3] * m[0] + t[4] * m[1] + t[5];
    double ex = px - m[2], ey = py - m[3];
    return sqrt(ex * ex + ey * ey);
}

/* :400-517.  returns 1 and fills out on success */
static int ransac_affine(const double *matches, size_t n, int method, int num_threads, orc_affine_result *out) {
    size_t min_sample = method == M_AFFINE ? 3 : 2;
    if (n < min_sample) return 0;
    size_t T = num_threads > 0 ? (size_t)num_threads : 1;
    size_t chunk = (RANSAC_ITERATIONS + T - 1) / T;
    size_t best_inliers = 0;
    double best_t[6] = {1, 0, 0, 0, 1, 0};
    unsigned char *best_mask = (unsigned char *)calloc(n, 1), *mask = (unsigned char *)malloc(n), *lmask = (unsigned char *)calloc(n, 1);
    double *sm = (double *)malloc(4 * min_sample * sizeof(double));
    for (size_t tid = 0; tid < T; tid++) {
        uint64_t state = 0xDEADBEEFCAFEBABEull + (uint64_t)tid * 0x9E3779B97F4A7C15ull;
        size_t local_best = 0;
        double local_t[6] = {1, 0, 0, 0, 1, 0};
        memset(lmask, 0, n);
        for (size_t it = 0; it < chunk; it++) {
            size_t sample[3], ns = 0;
            int attempts = 0;
            while (ns < min_sample && attempts < 20) {
                state ^= state << 13; state ^= state >> 7; state ^= state << 17;
                size_t idx = (size_t)(state % (uint64_t)n);
                int dup = 0;
                for (size_t q = 0; q < ns; q++) if (sample[q] == idx) dup = 1;
                if (!dup) sample[ns++] = idx;
                attempts++;
            }
            if (ns < min_sample) continue;
            for (size_t q = 0; q < ns; q++) memcpy(sm + 4 * q, matches + 4 * sample[q], 4 * sizeof(double));
            double tr[6];
            int ok = method == M_AFFINE ? orc_fit_affine(sm, ns, tr) : orc_fit_rigid(sm, ns, tr);
            if (!ok) continue;
            size_t cnt = 0;
            for (size_t i = 0; i < n; i++) { mask[i] = point_err(tr, matches + 4 * i) < RANSAC_INLIER_PX; cnt += mask[i]; }
            if (cnt > local_best) { local_best = cnt; memcpy(local_t, tr, sizeof tr); memcpy(lmask, mask, n); }
        }
        if (tid == 0 || local_best > best_inliers) {                   /* reduce_with: leftmost maximum */
            best_inliers = local_best; memcpy(best_t, local_t, sizeof best_t); memcpy(best_mask, lmask, n);
        }
    }
    int ok = 0;
    if (best_inliers >= MIN_MATCHES_RIGID && (double)best_inliers / (double)n >= MIN_INLIER_RATIO) {
        double *im = (double *)malloc(4 * n * sizeof(double));
        size_t ni = 0;
        for (size_t i = 0; i < n; i++) if (best_mask[i]) { memcpy(im + 4 * ni, matches + 4 * i, 4 * sizeof(double)); ni++; }
        double refined[6];
        int fit = method == M_AFFINE ? orc_fit_affine(im, ni, refined) : orc_fit_rigid(im, ni, refined);
        if (!fit) memcpy(refined, best_t, sizeof refined);
        double residual = 0.0;
        if (ni > 0) { double s = 0.0; for (size_t i = 0; i < ni; i++) s += point_err(refined, im + 4 * i); residual = s / (double)ni; }
        if (!(residual > MAX_RESIDUAL_PX)) {
            memcpy(out->t, refined, sizeof refined);
            out->matched_stars = n; out->inliers = best_inliers; out->residual_px = residual; out->method = method;
            ok = 1;
        }
        free(im);
    }
    free(best_mask); free(mask); free(lmask); free(sm);
    return ok;
}

/* :214-241 */
static int transform_sane(const orc_affine_result *r, size_t rows, size_t cols) {
    const double *t = r->t;
    if (fabs(t[2]) > (double)cols * MAX_OFFSET_FRACTION || fabs(t[5]) > (double)rows * MAX_OFFSET_FRACTION) return 0;
    double rot = fabs(atan2(t[3], t[0]) * (180.0 / 3.14159265358979323846));
    if (rot > MAX_ROTATION_DEG) return 0;
    double sx = sqrt(t[0] * t[0] + t[3] * t[3]), sy = sqrt(t[1] * t[1] + t[4] * t[4]);
    if (sx < MIN_SCALE || sx > MAX_SCALE || sy < MIN_SCALE || sy > MAX_SCALE) return 0;
    return 1;
}

/* the star-list half of align_channel_affine (:146-209); 1 = produced a star-based transform */
int orc_affine_from_stars(const double *ref_xy, size_t n_ref, const double *tgt_xy, size_t n_tgt, size_t rows, size_t cols,
                          int num_threads, orc_affine_result *out) {
    if (n_ref > MAX_STARS) n_ref = MAX_STARS;                            /* top_n_stars */
    if (n_tgt > MAX_STARS) n_tgt = MAX_STARS;
    if (n_ref < MIN_MATCHES_RIGID || n_tgt < MIN_MATCHES_RIGID) return 0;
    size_t nrt, ntt;
    tri_t *rt = build_triangles(ref_xy, n_ref, &nrt), *tt = build_triangles(tgt_xy, n_tgt, &ntt);
    int done = 0;
    if (nrt && ntt) {
        double *matches = (double *)malloc(4 * (n_ref < n_tgt ? n_ref : n_tgt) * sizeof(double) + 32);
        size_t nm = match_triangles(ref_xy, n_ref, tgt_xy, n_tgt, rt, nrt, tt, ntt, matches);
        if (nm >= MIN_MATCHES_RIGID) {
            if (nm >= MIN_MATCHES_AFFINE) {
                orc_affine_result r;
                if (ransac_affine(matches, nm, M_AFFINE, num_threads, &r) && transform_sane(&r, rows, cols)) { *out = r; done = 1; }
            }
            if (!done) {
                orc_affine_result r;
                if (ransac_affine(matches, nm, M_RIGID, num_threads, &r) && transform_sane(&r, rows, cols)) { *out = r; done = 1; }
            }
        }
        free(matches);
    }
    free(rt); free(tt);
    return done;
}

/* :129-212 + :243-270 */
void orc_align_channel_affine(const float *reference, const float *target, size_t rows, size_t cols, int num_threads,
                              orc_affine_result *out) {
    size_t len = rows * cols;
    float *rn = (float *)malloc(len * sizeof(float)), *tn = (float *)malloc(len * sizeof(float));
    orc_normalize_for_detection(reference, len, rn);
    orc_normalize_for_detection(target, len, tn);
    orc_star *rs = (orc_star *)malloc(MAX_STARS * sizeof(orc_star)), *ts = (orc_star *)malloc(MAX_STARS * sizeof(orc_star));
    size_t tot;
    double bm, bs;
    size_t nr = orc_detect_stars(rn, rows, cols, DETECTION_SIGMA, rs, MAX_STARS, &tot, &bm, &bs);
    size_t nt = orc_detect_stars(tn, rows, cols, DETECTION_SIGMA, ts, MAX_STARS, &tot, &bm, &bs);
    double *rxy = (double *)malloc(2 * MAX_STARS * sizeof(double)), *txy = (double *)malloc(2 * MAX_STARS * sizeof(double));
    for (size_t i = 0; i < nr; i++) { rxy[2 * i] = rs[i].x; rxy[2 * i + 1] = rs[i].y; }
    for (size_t i = 0; i < nt; i++) { txy[2 * i] = ts[i].x; txy[2 * i + 1] = ts[i].y; }
    if (!orc_affine_from_stars(rxy, nr, txy, nt, rows, cols, num_threads, out)) {
        double dx, dy, conf;                                             /* fallback_phase_correlation */
        orc_phase_correlate(reference, rows, cols, target, rows, cols, &dx, &dy, &conf);
        memset(out, 0, sizeof *out);
        out->t[0] = 1.0; out->t[4] = 1.0;
        if (fabs(dx) > (double)cols * MAX_OFFSET_FRACTION || fabs(dy) > (double)rows * MAX_OFFSET_FRACTION || conf < 1.5) {
            out->method = M_IDENTITY;
        } else {
            out->t[2] = dx; out->t[5] = dy; out->method = M_PC;
        }
    }
    free(rn); free(tn); free(rs); free(ts); free(rxy); free(txy);
}
