// Star-based affine / rigid registration on gfx950 + host.
//
// Replaces core/alignment/affine.rs: align_channel_affine (:129-212), check_transform_sanity
// (:214-241), fallback_phase_correlation (:243-270), top_n_stars (:272-277), build_triangles
// (:279-318), match_triangles (:320-384), sort_triangle_vertices (:386-398), ransac_affine
// (:400-517), fit_affine / solve_3x3_ls / solve_3x3 (:519-595), fit_rigid (:597-642),
// compute_residual (:644-656).
//
// Split: the per-pixel work (percentile normalisation, tile background, threshold, labelling:
// csrc/detect.hip; phase-correlation fallback: csrc/phase_corr.hip) runs on the GPU; what is left
// is scalar f64 geometry over <= 120 stars (<= 34 220 triangles per image, 2000 RANSAC draws),
// which the reference also runs as plain CPU code and which stays on the host here.
//
// Two places where the reference's answer is not a function of its inputs are pinned:
//   * vote pairs are iterated out of a std::HashMap and stable-sorted by votes only (:351-360):
//     ties come out in random order.  Here: votes descending, then (ref index, tgt index) ascending.
//   * RANSAC splits its 2000 iterations over rayon::current_num_threads() workers, each with its
//     own xorshift seed (:410-416): the result depends on the host's core count.  Here the worker
//     count is an explicit argument of the ABI.
#include "ab_common.hpp"

#include <algorithm>
#include <array>
#include <cmath>

int ab_detect_stars_device(ab_ctx *ctx, const float *img, int64_t rows, int64_t cols, int64_t ld, double sigma_threshold,
                           std::vector<ab_detected_star> *stars, double *bg_median_out, double *bg_sigma_out);
int ab_normalize_for_detection_device(ab_ctx *ctx, const float *img, int64_t len, float *out, int *cloned);
int ab_phase_correlate_device(ab_ctx *ctx, const float *ref, int64_t ref_rows, int64_t ref_cols, int64_t ref_ld, const float *tgt,
                              int64_t tgt_rows, int64_t tgt_cols, int64_t tgt_ld, double *dx, double *dy, double *confidence);

namespace {

constexpr size_t kMaxStars = 120;            // affine.rs:8-22
constexpr double kTriangleTolerance = 0.02;
constexpr size_t kMinMatchesAffine = 6, kMinMatchesRigid = 4;
constexpr size_t kRansacIterations = 2000;
constexpr double kRansacInlierPx = 3.0, kDetectionSigma = 3.5, kMinTriangleSide = 15.0;
constexpr uint32_t kMinVotes = 1;
constexpr double kMinInlierRatio = 0.20, kMaxResidualPx = 5.0, kMaxOffsetFraction = 0.40, kMaxRotationDeg = 30.0;
constexpr double kMinScale = 0.70, kMaxScale = 1.40;

using Pt = std::array<double, 2>;
using Match = std::array<double, 4>;  // rx, ry, tx, ty
using Xf = std::array<double, 6>;     // a, b, tx, c, d, ty

struct Tri {
    size_t idx[3];
    double ratio_mid, ratio_long;
};

double dist(const Pt &a, const Pt &b) {  // :658-661
    const double dx = a[0] - b[0], dy = a[1] - b[1];
    return std::sqrt(dx * dx + dy * dy);
}

std::vector<Tri> build_triangles(const std::vector<Pt> &s) {  // :279-318
    std::vector<Tri> tris;
    const size_t n = s.size();
    if (n < 3) return tris;
    const size_t limit = std::min<size_t>(n, 60);
    for (size_t i = 0; i < limit; ++i)
        for (size_t j = i + 1; j < limit; ++j)
            for (size_t k = j + 1; k < limit; ++k) {
                std::array<double, 3> sides = {dist(s[i], s[j]), dist(s[j], s[k]), dist(s[i], s[k])};
                std::stable_sort(sides.begin(), sides.end());
                if (sides[0] < kMinTriangleSide) continue;
                tris.push_back({{i, j, k}, sides[1] / sides[0], sides[2] / sides[0]});
            }
    return tris;
}

std::array<size_t, 3> sort_triangle_vertices(const std::vector<Pt> &s, const size_t idx[3]) {  // :386-398
    std::array<std::pair<size_t, double>, 3> v = {{{idx[0], dist(s[idx[1]], s[idx[2]])},
                                                   {idx[1], dist(s[idx[0]], s[idx[2]])},
                                                   {idx[2], dist(s[idx[0]], s[idx[1]])}}};
    std::stable_sort(v.begin(), v.end(), [](const auto &a, const auto &b) { return a.second < b.second; });
    return {v[0].first, v[1].first, v[2].first};
}

std::vector<Match> match_triangles(const std::vector<Pt> &rs, const std::vector<Pt> &ts, const std::vector<Tri> &rt,
                                   const std::vector<Tri> &tt) {  // :320-384
    const size_t nr = rs.size(), nt = ts.size();
    std::vector<uint32_t> votes(nr * nt, 0);
    // The reference compares every ref triangle with every tgt triangle (up to 34 220^2 pairs).  Votes are
    // integer counts, so the visiting order is free: sort the tgt triangles by ratio_mid and only visit the
    // window that can pass `|d_mid| <= 0.02` (a hair wider than the tolerance; the exact test decides).
    std::vector<size_t> order(tt.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return tt[a].ratio_mid < tt[b].ratio_mid; });
    std::vector<double> mids(tt.size());
    std::vector<std::array<size_t, 3>> tverts(tt.size());
    for (size_t i = 0; i < order.size(); ++i) {
        mids[i] = tt[order[i]].ratio_mid;
        tverts[i] = sort_triangle_vertices(ts, tt[order[i]].idx);
    }
    for (const Tri &a : rt) {
        const double lo = a.ratio_mid - kTriangleTolerance * 1.000001 - 1e-12, hi = a.ratio_mid + kTriangleTolerance * 1.000001 + 1e-12;
        const size_t i0 = std::lower_bound(mids.begin(), mids.end(), lo) - mids.begin();
        bool have_ra = false;
        std::array<size_t, 3> ra{};
        for (size_t i = i0; i < mids.size() && mids[i] <= hi; ++i) {
            const Tri &b = tt[order[i]];
            if (std::fabs(a.ratio_mid - b.ratio_mid) > kTriangleTolerance || std::fabs(a.ratio_long - b.ratio_long) > kTriangleTolerance)
                continue;
            if (!have_ra) {
                ra = sort_triangle_vertices(rs, a.idx);
                have_ra = true;
            }
            for (int p = 0; p < 3; ++p) votes[ra[p] * nt + tverts[i][p]] += 1;
        }
    }
    struct Pair {
        size_t ri, ti;
        uint32_t v;
    };
    std::vector<Pair> pairs;
    for (size_t r = 0; r < nr; ++r)
        for (size_t t = 0; t < nt; ++t)
            if (votes[r * nt + t]) pairs.push_back({r, t, votes[r * nt + t]});
    std::sort(pairs.begin(), pairs.end(), [](const Pair &a, const Pair &b) {
        if (a.v != b.v) return a.v > b.v;
        if (a.ri != b.ri) return a.ri < b.ri;
        return a.ti < b.ti;
    });
    std::vector<char> used_r(nr, 0), used_t(nt, 0);
    std::vector<Match> out;
    for (const Pair &p : pairs) {
        if (p.v < kMinVotes) break;
        if (used_r[p.ri] || used_t[p.ti]) continue;
        used_r[p.ri] = used_t[p.ti] = 1;
        out.push_back({rs[p.ri][0], rs[p.ri][1], ts[p.ti][0], ts[p.ti][1]});
    }
    return out;
}

bool solve_3x3(const double a[3][3], const double b[3], double x[3]) {  // :556-595
    const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                       a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    if (std::fabs(det) < 1e-12) return false;
    const double id = 1.0 / det;
    const double inv[3][3] = {
        {(a[1][1] * a[2][2] - a[1][2] * a[2][1]) * id, (a[0][2] * a[2][1] - a[0][1] * a[2][2]) * id, (a[0][1] * a[1][2] - a[0][2] * a[1][1]) * id},
        {(a[1][2] * a[2][0] - a[1][0] * a[2][2]) * id, (a[0][0] * a[2][2] - a[0][2] * a[2][0]) * id, (a[0][2] * a[1][0] - a[0][0] * a[1][2]) * id},
        {(a[1][0] * a[2][1] - a[1][1] * a[2][0]) * id, (a[0][1] * a[2][0] - a[0][0] * a[2][1]) * id, (a[0][0] * a[1][1] - a[0][1] * a[1][0]) * id}};
    for (int i = 0; i < 3; ++i) x[i] = inv[i][0] * b[0] + inv[i][1] * b[1] + inv[i][2] * b[2];
    return true;
}

bool solve_3x3_ls(const Match *m, size_t n, bool solve_x, double out[3]) {  // :538-554
    double ata[3][3] = {{0}}, atb[3] = {0};
    for (size_t k = 0; k < n; ++k) {
        const double target = solve_x ? m[k][2] : m[k][3];
        const double row[3] = {m[k][0], m[k][1], 1.0};
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) ata[i][j] += row[i] * row[j];
            atb[i] += row[i] * target;
        }
    }
    return solve_3x3(ata, atb, out);
}

bool fit_affine(const Match *m, size_t n, Xf &t) {  // :519-536
    if (n < 3) return false;
    double ab[3], cd[3];
    if (!solve_3x3_ls(m, n, true, ab) || !solve_3x3_ls(m, n, false, cd)) return false;
    t = {ab[0], ab[1], ab[2], cd[0], cd[1], cd[2]};
    return true;
}

bool fit_rigid(const Match *m, size_t n, Xf &t) {  // :597-642
    if (n < 2) return false;
    double rcx = 0, rcy = 0, tcx = 0, tcy = 0;
    for (size_t k = 0; k < n; ++k) {
        rcx += m[k][0];
        rcy += m[k][1];
        tcx += m[k][2];
        tcy += m[k][3];
    }
    const double nf = (double)n;
    rcx /= nf;
    rcy /= nf;
    tcx /= nf;
    tcy /= nf;
    double num = 0, den = 0;
    for (size_t k = 0; k < n; ++k) {
        const double drx = m[k][0] - rcx, dry = m[k][1] - rcy, dtx = m[k][2] - tcx, dty = m[k][3] - tcy;
        num += drx * dty - dry * dtx;
        den += drx * dtx + dry * dty;
    }
    const double theta = std::atan2(num, den), c = std::cos(theta), s = std::sin(theta);
    t = {c, -s, tcx - c * rcx + s * rcy, s, c, tcy - s * rcx - c * rcy};
    return true;
}

double point_err(const Xf &t, const Match &m) {
    const double px = t[0] * m[0] + t[1] * m[1] + t[2], py = t[3] * m[0] + t[4] * m[1] + t[5];
    const double ex = px - m[2], ey = py - m[3];
    return std::sqrt(ex * ex + ey * ey);
}

enum { kAffine = 0, kRigid = 1, kPhaseCorr = 2, kIdentity = 3 };

bool ransac(const std::vector<Match> &matches, int method, int num_threads, ab_affine_align_result *out) {  // :400-517
    const size_t n = matches.size(), min_sample = method == kAffine ? 3 : 2;
    if (n < min_sample) return false;
    const size_t T = (size_t)std::max(num_threads, 1), chunk = (kRansacIterations + T - 1) / T;
    size_t best_inliers = 0;
    Xf best_t = {1, 0, 0, 0, 1, 0};
    std::vector<char> best_mask(n, 0), mask(n), lmask(n);
    for (size_t tid = 0; tid < T; ++tid) {
        uint64_t state = 0xDEADBEEFCAFEBABEull + (uint64_t)tid * 0x9E3779B97F4A7C15ull;
        size_t local_best = 0;
        Xf local_t = {1, 0, 0, 0, 1, 0};
        std::fill(lmask.begin(), lmask.end(), 0);
        for (size_t it = 0; it < chunk; ++it) {
            size_t sample[3], ns = 0;
            for (int attempts = 0; ns < min_sample && attempts < 20; ++attempts) {
                state ^= state << 13;
                state ^= state >> 7;
                state ^= state << 17;
                const size_t idx = (size_t)(state % (uint64_t)n);
                bool dup = false;
                for (size_t q = 0; q < ns; ++q) dup |= sample[q] == idx;
                if (!dup) sample[ns++] = idx;
            }
            if (ns < min_sample) continue;
            Match sm[3];
            for (size_t q = 0; q < ns; ++q) sm[q] = matches[sample[q]];
            Xf tr;
            if (!(method == kAffine ? fit_affine(sm, ns, tr) : fit_rigid(sm, ns, tr))) continue;
            size_t cnt = 0;
            for (size_t i = 0; i < n; ++i) {
                mask[i] = point_err(tr, matches[i]) < kRansacInlierPx;
                cnt += mask[i];
            }
            if (cnt > local_best) {
                local_best = cnt;
                local_t = tr;
                lmask = mask;
            }
        }
        if (tid == 0 || local_best > best_inliers) {  // reduce_with(|a, b| if b.0 > a.0 { b } else { a }): leftmost maximum
            best_inliers = local_best;
            best_t = local_t;
            best_mask = lmask;
        }
    }
    if (best_inliers < kMinMatchesRigid) return false;
    if ((double)best_inliers / (double)n < kMinInlierRatio) return false;
    std::vector<Match> in;
    for (size_t i = 0; i < n; ++i)
        if (best_mask[i]) in.push_back(matches[i]);
    Xf refined;
    if (!(method == kAffine ? fit_affine(in.data(), in.size(), refined) : fit_rigid(in.data(), in.size(), refined))) refined = best_t;
    double residual = 0.0;
    if (!in.empty()) {
        double s = 0.0;
        for (const Match &m : in) s += point_err(refined, m);
        residual = s / (double)in.size();
    }
    if (residual > kMaxResidualPx) return false;
    for (int i = 0; i < 6; ++i) out->transform[i] = refined[i];
    out->matched_stars = n;
    out->inliers = best_inliers;
    out->residual_px = residual;
    out->method = method;
    return true;
}

bool transform_sane(const ab_affine_align_result &r, int64_t rows, int64_t cols) {  // :214-241
    const double *t = r.transform;
    if (std::fabs(t[2]) > (double)cols * kMaxOffsetFraction || std::fabs(t[5]) > (double)rows * kMaxOffsetFraction) return false;
    const double rot = std::fabs(std::atan2(t[3], t[0]) * (180.0 / 3.14159265358979323846));
    if (rot > kMaxRotationDeg) return false;
    const double sx = std::sqrt(t[0] * t[0] + t[3] * t[3]), sy = std::sqrt(t[1] * t[1] + t[4] * t[4]);
    return !(sx < kMinScale || sx > kMaxScale || sy < kMinScale || sy > kMaxScale);
}

// the star-list half of align_channel_affine (:146-209)
bool affine_from_stars(std::vector<Pt> rs, std::vector<Pt> ts, int64_t rows, int64_t cols, int num_threads,
                       ab_affine_align_result *out) {
    if (rs.size() > kMaxStars) rs.resize(kMaxStars);  // top_n_stars
    if (ts.size() > kMaxStars) ts.resize(kMaxStars);
    if (rs.size() < kMinMatchesRigid || ts.size() < kMinMatchesRigid) return false;
    const auto rt = build_triangles(rs), tt = build_triangles(ts);
    if (rt.empty() || tt.empty()) return false;
    const auto matches = match_triangles(rs, ts, rt, tt);
    if (matches.size() < kMinMatchesRigid) return false;
    ab_affine_align_result r;
    if (matches.size() >= kMinMatchesAffine && ransac(matches, kAffine, num_threads, &r) && transform_sane(r, rows, cols)) {
        *out = r;
        return true;
    }
    if (ransac(matches, kRigid, num_threads, &r) && transform_sane(r, rows, cols)) {
        *out = r;
        return true;
    }
    return false;
}

}  // namespace

// align_channel_affine (affine.rs:129-212) on device planes of equal dims (row strides allowed)
int ab_align_channel_affine_device(ab_ctx *ctx, const float *ref, const float *tgt, int64_t rows, int64_t cols, int num_threads,
                                   ab_affine_align_result *out) {
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const int64_t len = rows * cols;
    float *norm = nullptr;
    AB_HIP(ctx, hipMalloc((void **)&norm, (size_t)len * sizeof(float)));
    std::vector<ab_detected_star> rstars, tstars;
    double m, s;
    int cloned = 0;
    int rc = ab_normalize_for_detection_device(ctx, ref, len, norm, &cloned);
    if (rc == AB_OK) rc = ab_detect_stars_device(ctx, norm, rows, cols, cols, kDetectionSigma, &rstars, &m, &s);
    if (rc == AB_OK) rc = ab_normalize_for_detection_device(ctx, tgt, len, norm, &cloned);
    if (rc == AB_OK) rc = ab_detect_stars_device(ctx, norm, rows, cols, cols, kDetectionSigma, &tstars, &m, &s);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(norm);
    if (rc != AB_OK) return rc;
    std::vector<Pt> rs, ts;
    for (const auto &st : rstars) rs.push_back({st.x, st.y});
    for (const auto &st : tstars) ts.push_back({st.x, st.y});
    if (affine_from_stars(rs, ts, rows, cols, num_threads, out)) return AB_OK;
    // fallback_phase_correlation (:243-270) on the ORIGINAL planes
    double dx, dy, conf;
    AB_TRY(ab_phase_correlate_device(ctx, ref, rows, cols, cols, tgt, rows, cols, cols, &dx, &dy, &conf));
    memset(out, 0, sizeof *out);
    out->transform[0] = 1.0;
    out->transform[4] = 1.0;
    if (std::fabs(dx) > (double)cols * kMaxOffsetFraction || std::fabs(dy) > (double)rows * kMaxOffsetFraction || conf < 1.5) {
        out->method = kIdentity;
    } else {
        out->transform[2] = dx;
        out->transform[5] = dy;
        out->method = kPhaseCorr;
    }
    return AB_OK;
}

extern "C" {

int ab_align_channel_affine(ab_ctx *ctx, const ab_plane *reference, const ab_plane *target, int num_threads,
                            ab_affine_align_result *out) {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, reference && target && out, "null argument");
    AB_CHECK(ctx, reference->rows == target->rows && reference->cols == target->cols,
             "align_channel_affine takes two planes of the same dims (%lldx%lld vs %lldx%lld)", (long long)reference->rows,
             (long long)reference->cols, (long long)target->rows, (long long)target->cols);
    StagedPlane r, t;
    AB_TRY(ab_stage_in(ctx, reference, &r));
    int rc = ab_stage_in(ctx, target, &t);
    if (rc == AB_OK) {
        rc = ab_align_channel_affine_device(ctx, r.dptr, t.dptr, r.rows, r.cols, num_threads, out);
        ab_stage_release(ctx, &t);
    }
    ab_stage_release(ctx, &r);
    return rc;
}

// the host geometry alone, on given centroids (x, y pairs): returns AB_OK and *found = 0/1
int ab_affine_from_stars(const double *ref_xy, size_t n_ref, const double *tgt_xy, size_t n_tgt, int64_t rows, int64_t cols,
                         int num_threads, ab_affine_align_result *out, int *found) {
    if ((!ref_xy && n_ref) || (!tgt_xy && n_tgt) || !out || !found) return AB_ERR_INVALID;
    std::vector<Pt> rs(n_ref), ts(n_tgt);
    for (size_t i = 0; i < n_ref; ++i) rs[i] = {ref_xy[2 * i], ref_xy[2 * i + 1]};
    for (size_t i = 0; i < n_tgt; ++i) ts[i] = {tgt_xy[2 * i], tgt_xy[2 * i + 1]};
    *found = affine_from_stars(rs, ts, rows, cols, num_threads, out) ? 1 : 0;
    return AB_OK;
}

}  // extern "C"
