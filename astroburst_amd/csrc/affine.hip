// Star-based affine / rigid registration on gfx950 + host.
//
// Replaces core/alignment/affine.rs: align_channel_affine (:129-212), check_transform_sanity
// (:214-241), fallback_phase_correlation (:243-270), top_n_stars (:272-277), build_triangles
// (:279-318), match_triangles (:320-384), sort_triangle_vertices (:386-398), ransac_affine
// (:400-517), fit_affine / solve_3x3_ls / solve_3x3 (:519-595), fit_rigid (:597-642),
// compute_residual (:644-656).
//
// Split: the per-pixel work (percentile normalisation, tile background, threshold, labelling, moments:
// csrc/detect.hip; phase-correlation fallback: csrc/phase_corr.hip) and the triangle table + vote matrix
// (<= 34 220 triangles per image, up to 1.2e9 pair tests) run on the GPU; RANSAC (2000 draws over <= 60
// matches) and the final fits are scalar f64 geometry that stays on the host.  ab_register_frames /
// ab_align_pairs_affine spread the targets of one reference over host worker threads, one HIP stream each.
//
// Two places where the reference's answer is not a function of its inputs are pinned:
//   * vote pairs are iterated out of a std::HashMap and stable-sorted by votes only (:351-360):
//     ties come out in random order.  Here: votes descending, then (ref index, tgt index) ascending.
//   * RANSAC splits its 2000 iterations over rayon::current_num_threads() workers, each with its
//     own xorshift seed (:410-416): the result depends on the host's core count.  Here the worker
//     count is an explicit argument of the ABI.
#include "ab_common.hpp"

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>

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
constexpr size_t kHostVoteDim = 64;  // >= build_triangles' 60-star limit
constexpr size_t kVotingStars = 60;  // build_triangles' `limit` (:285): only the first 60 stars of a list form triangles and collect votes

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

// stable sort of three elements (what slice::sort_by does for len 3: insertion sort), without std::stable_sort's
// temporary buffer -- it runs 10^5 times per frame pair
template <class A, class Less>
void sort3(A &v, Less lt) {
    if (lt(v[1], v[0])) std::swap(v[0], v[1]);
    if (lt(v[2], v[1])) {
        std::swap(v[1], v[2]);
        if (lt(v[1], v[0])) std::swap(v[0], v[1]);
    }
}

std::vector<Tri> build_triangles(const std::vector<Pt> &s) {  // :279-318
    std::vector<Tri> tris;
    const size_t n = s.size();
    if (n < 3) return tris;
    const size_t limit = std::min<size_t>(n, 60);
    tris.reserve(limit * (limit - 1) * (limit - 2) / 6);
    for (size_t i = 0; i < limit; ++i)
        for (size_t j = i + 1; j < limit; ++j)
            for (size_t k = j + 1; k < limit; ++k) {
                std::array<double, 3> sides = {dist(s[i], s[j]), dist(s[j], s[k]), dist(s[i], s[k])};
                sort3(sides, [](double a, double b) { return a < b; });
                if (sides[0] < kMinTriangleSide) continue;
                tris.push_back({{i, j, k}, sides[1] / sides[0], sides[2] / sides[0]});
            }
    return tris;
}

std::array<size_t, 3> sort_triangle_vertices(const std::vector<Pt> &s, const size_t idx[3]) {  // :386-398
    std::array<std::pair<size_t, double>, 3> v = {{{idx[0], dist(s[idx[1]], s[idx[2]])},
                                                   {idx[1], dist(s[idx[0]], s[idx[2]])},
                                                   {idx[2], dist(s[idx[0]], s[idx[1]])}}};
    sort3(v, [](const auto &a, const auto &b) { return a.second < b.second; });
    return {v[0].first, v[1].first, v[2].first};
}

// votes -> one-to-one matches (:351-384): pairs by votes descending (ties: ref index, then tgt index -- pinned,
// see the file header), greedily keeping pairs whose two stars are both still free
std::vector<Match> matches_from_votes(const std::vector<Pt> &rs, const std::vector<Pt> &ts, const uint32_t *votes, size_t stride) {
    const size_t nr = rs.size(), nt = ts.size();
    // one u64 per pair, ordered like (votes descending, ref index, tgt index): ~votes in the high word, the indices below
    std::vector<uint64_t> pairs;
    pairs.reserve(stride * stride);
    for (size_t r = 0; r < nr && r < stride; ++r)  // votes is stride x stride; only the first <= 60 stars of a list vote
        for (size_t t = 0; t < nt && t < stride; ++t) {
            const uint32_t v = votes[r * stride + t];
            if (v >= kMinVotes) pairs.push_back(((uint64_t)(0xffffffffu - v) << 32) | ((uint64_t)r << 16) | (uint64_t)t);  // (the greedy pass stops at the first pair below kMinVotes)
        }
    // The greedy pass ends as soon as every star of the shorter list is taken, which the few hundred best pairs achieve: they
    // are split off and ordered first, the thousands of one- and two-vote pairs behind them only if the pass gets that far
    // (sorting all ~3600 pairs was 0.1 ms per frame).
    size_t sorted_upto = pairs.size();
    if (pairs.size() > 1024) {
        sorted_upto = 512;
        std::nth_element(pairs.begin(), pairs.begin() + sorted_upto, pairs.end());
    }
    std::sort(pairs.begin(), pairs.begin() + sorted_upto);
    std::vector<char> used_r(nr, 0), used_t(nt, 0);
    std::vector<Match> out;
    const size_t full = std::min(std::min(nr, nt), kVotingStars);  // every voting star of the shorter list taken: nothing further can be accepted
    for (size_t i = 0; i < pairs.size() && out.size() < full; ++i) {
        if (i == sorted_upto) {  // the rest: only pairs whose two stars are both still free can ever be accepted -- usually a handful
            const auto dead = [&](uint64_t q) { return used_r[(size_t)((q >> 16) & 0xffffu)] || used_t[(size_t)(q & 0xffffu)]; };
            pairs.erase(std::remove_if(pairs.begin() + sorted_upto, pairs.end(), dead), pairs.end());
            std::sort(pairs.begin() + sorted_upto, pairs.end());
            sorted_upto = pairs.size();
            if (i >= pairs.size()) break;
        }
        const uint64_t p = pairs[i];
        const size_t ri = (size_t)((p >> 16) & 0xffffu), ti = (size_t)(p & 0xffffu);
        if (used_r[ri] || used_t[ti]) continue;
        used_r[ri] = used_t[ti] = 1;
        out.push_back({rs[ri][0], rs[ri][1], ts[ti][0], ts[ti][1]});
    }
    return out;
}

std::vector<Match> match_triangles(const std::vector<Pt> &rs, const std::vector<Pt> &ts, const std::vector<Tri> &rt,
                                   const std::vector<Tri> &tt) {  // :320-384
    const size_t nr = rs.size(), nt = ts.size();
    std::vector<uint32_t> votes(nr * nt, 0);
    // The reference compares every ref triangle with every tgt triangle (up to 34 220^2 pairs).  Votes are
    // integer counts, so the visiting order is free: sort the tgt triangles by ratio_mid and only visit the
    // window that can pass `|d_mid| <= 0.02` (a hair wider than the tolerance; the exact test decides).
    // Bucket the tgt triangles on a (ratio_mid, ratio_long) grid whose cells are a hair wider than the tolerance:
    // two triangles within tolerance on both ratios sit in the same or in adjacent cells, so the 3 x 3
    // neighbourhood is a superset of the matches and the reference's exact test decides inside it.
    const double cell = kTriangleTolerance * 1.0001;
    auto key_of = [&](double mid, double lng) { return ((uint64_t)(mid / cell) << 24) | (uint64_t)std::min(lng / cell, 16777215.0); };
    std::vector<std::pair<uint64_t, uint32_t>> keyed(tt.size());
    for (size_t i = 0; i < tt.size(); ++i) keyed[i] = {key_of(tt[i].ratio_mid, tt[i].ratio_long), (uint32_t)i};
    std::sort(keyed.begin(), keyed.end());
    std::vector<std::array<uint8_t, 3>> tverts(tt.size());
    for (size_t i = 0; i < tt.size(); ++i) {
        const auto v = sort_triangle_vertices(ts, tt[i].idx);
        tverts[i] = {(uint8_t)v[0], (uint8_t)v[1], (uint8_t)v[2]};
    }
    for (const Tri &a : rt) {
        const uint64_t cm = (uint64_t)(a.ratio_mid / cell), cl = (uint64_t)std::min(a.ratio_long / cell, 16777215.0);
        bool have_ra = false;
        std::array<size_t, 3> ra{};
        for (uint64_t m = cm ? cm - 1 : 0; m <= cm + 1; ++m) {
            const uint64_t k0 = (m << 24) | (cl ? cl - 1 : 0), k1 = (m << 24) | std::min<uint64_t>(cl + 1, 16777215);
            auto it = std::lower_bound(keyed.begin(), keyed.end(), std::make_pair(k0, (uint32_t)0));
            for (; it != keyed.end() && it->first <= k1; ++it) {
                const Tri &b = tt[it->second];
                if (std::fabs(a.ratio_mid - b.ratio_mid) > kTriangleTolerance || std::fabs(a.ratio_long - b.ratio_long) > kTriangleTolerance)
                    continue;
                if (!have_ra) {
                    ra = sort_triangle_vertices(rs, a.idx);
                    have_ra = true;
                }
                for (int p = 0; p < 3; ++p) votes[ra[p] * nt + tverts[it->second][p]] += 1;
            }
        }
    }
    std::vector<uint32_t> sq(kHostVoteDim * kHostVoteDim, 0);
    for (size_t r = 0; r < nr && r < kHostVoteDim; ++r)
        for (size_t t = 0; t < nt && t < kHostVoteDim; ++t) sq[r * kHostVoteDim + t] = votes[r * nt + t];
    return matches_from_votes(rs, ts, sq.data(), kHostVoteDim);
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
    std::vector<char> best_mask(n, 0), lmask(n);
    static_assert(kRansacInlierPx == 3.0, "the squared inlier test above is exact for 3.0; check it for another threshold");
    std::vector<double> mx(n), my(n), mtx(n), mty(n);
    for (size_t i = 0; i < n; ++i) {
        mx[i] = matches[i][0];
        my[i] = matches[i][1];
        mtx[i] = matches[i][2];
        mty[i] = matches[i][3];
    }
    for (size_t tid = 0; tid < T; ++tid) {
        uint64_t state = 0xDEADBEEFCAFEBABEull + (uint64_t)tid * 0x9E3779B97F4A7C15ull;
        size_t local_best = 0;
        Xf local_t = {1, 0, 0, 0, 1, 0};
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
            // inliers of this draw: point_err(tr, m) < 3  <=>  ex^2 + ey^2 < 9 exactly (9 is a square, sqrt is monotone and
            // correctly rounded, and the double below 9 has its root below 3), so the count needs no square roots and no mask
            size_t cnt = 0;
            for (size_t i = 0; i < n; ++i) {
                const double px = tr[0] * mx[i] + tr[1] * my[i] + tr[2], py = tr[3] * mx[i] + tr[4] * my[i] + tr[5];
                const double ex = px - mtx[i], ey = py - mty[i];
                cnt += (ex * ex + ey * ey < kRansacInlierPx * kRansacInlierPx) ? 1 : 0;
            }
            if (cnt > local_best) {
                local_best = cnt;
                local_t = tr;
            }
        }
        for (size_t i = 0; i < n; ++i) lmask[i] = local_best ? (point_err(local_t, matches[i]) < kRansacInlierPx) : 0;  // the winning draw's mask
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

// matches -> transform (:178-209): affine RANSAC, then rigid, each followed by the sanity check
bool transform_from_matches(const std::vector<Match> &matches, int64_t rows, int64_t cols, int num_threads, ab_affine_align_result *out) {
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

// the star-list half of align_channel_affine (:146-209)
bool affine_from_stars(std::vector<Pt> rs, std::vector<Pt> ts, int64_t rows, int64_t cols, int num_threads,
                       ab_affine_align_result *out) {
    if (rs.size() > kMaxStars) rs.resize(kMaxStars);  // top_n_stars
    if (ts.size() > kMaxStars) ts.resize(kMaxStars);
    if (rs.size() < kMinMatchesRigid || ts.size() < kMinMatchesRigid) return false;
    const auto rt = build_triangles(rs), tt = build_triangles(ts);
    if (rt.empty() || tt.empty()) return false;
    return transform_from_matches(match_triangles(rs, ts, rt, tt), rows, cols, num_threads, out);
}

// ---- GPU triangle matcher -------------------------------------------------------------------------------
// build_triangles + match_triangles compare up to 34 220 x 34 220 triangle pairs per frame pair: 1.2e9 pair tests of
// four f64 ops each.  On the host that is tens of milliseconds even with bucketing; here the triangles of <= 60
// stars are built by one thread per (i, j, k) and every ref triangle is tested against every tgt triangle staged
// through LDS (~0.2 ms).  Side lengths, ratios and the tolerance tests are the same correctly-rounded f64
// operations as the host path, and votes are integers, so the vote matrix is identical to the host's.
struct DTri {
    double mid, lng;
    uint32_t verts;  // sort_triangle_vertices order: v0 | v1 << 8 | v2 << 16
    uint32_t bin;    // tri_bin(mid), kept so the bucketing passes need not redo the f64 division
};

constexpr int kTriLimit = 60;            // build_triangles' `limit` (:285)
constexpr int kMaxTris = 34220;          // C(60, 3)
constexpr int kVoteDim = 64;             // vote matrix stride (>= kTriLimit)
constexpr int kTriBins = 4096;           // ratio_mid buckets of the tgt table (cell = tolerance * 1.0001)

__host__ __device__ __forceinline__ int tri_bin(double mid) {  // ratio_mid >= 1; monotone, so |d_mid| <= tol => |d_bin| <= 1
    const double b = (mid - 1.0) / (kTriangleTolerance * 1.0001);
    return b >= (double)(kTriBins - 1) ? kTriBins - 1 : (int)b;
}

__device__ __forceinline__ double ddist(double ax, double ay, double bx, double by) {
    const double dx = ax - bx, dy = ay - by;
    return sqrt(dx * dx + dy * dy);
}

struct StarXY {
    double xy[kTriLimit * 2];
};

// (round 4: 256-thread blocks -- a 1024-thread workgroup with 16 KB of LDS waited for a CU with sixteen free wave slots inside a
// registration batch: 157 us on average for a kernel that takes 11 alone)
constexpr int kTriBlock = 256;
__device__ __forceinline__ void tri_build_body(const double *__restrict__ xy, int limit, DTri *__restrict__ out, unsigned int *count,
                                                         unsigned int *__restrict__ bin_hist /* nullable; zero on entry */,
                                                         unsigned int *__restrict__ votes_to_clear /* nullable: the vote matrices of the frame */,
                                                         int vote_words) {
    // (the vote kernel of this frame runs later on the same stream: clearing its matrices here saves a fill command per frame)
    if (votes_to_clear && (int)(blockIdx.x * kTriBlock + threadIdx.x) < vote_words) votes_to_clear[blockIdx.x * kTriBlock + threadIdx.x] = 0;
    __shared__ unsigned int lhist[kTriBins];  // this block's share of the bucket histogram: one global atomic per touched bucket
    if (bin_hist) {
        for (int b = threadIdx.x; b < kTriBins; b += kTriBlock) lhist[b] = 0;
        __syncthreads();
    }
    const int t = blockIdx.x * kTriBlock + threadIdx.x;
    const int lim = limit >= 3 ? limit : 3;  // (a frame of a group without enough stars: no triangle, no division by zero)
    const int i = t / (lim * lim), j = (t / lim) % lim, k = t % lim;
    bool ok = limit >= 3 && i < limit && i < j && j < k;
    DTri tri = {0.0, 0.0, 0u, 0u};
    if (ok) {
        const double xi = xy[2 * i], yi = xy[2 * i + 1], xj = xy[2 * j], yj = xy[2 * j + 1], xk = xy[2 * k], yk = xy[2 * k + 1];
        const double dij = ddist(xi, yi, xj, yj), djk = ddist(xj, yj, xk, yk), dik = ddist(xi, yi, xk, yk);
        double s0 = dij, s1 = djk, s2 = dik;  // :295-297, then a stable sort of three
        if (s1 < s0) { const double q = s0; s0 = s1; s1 = q; }
        if (s2 < s1) {
            const double q = s1; s1 = s2; s2 = q;
            if (s1 < s0) { const double q2 = s0; s0 = s1; s1 = q2; }
        }
        ok = !(s0 < kMinTriangleSide);
        tri.mid = s1 / s0;
        tri.lng = s2 / s0;
        // sort_triangle_vertices (:386-398): vertices by the length of the opposite side, stable
        int v0 = i, v1 = j, v2 = k;
        double o0 = djk, o1 = dik, o2 = dij;
        if (o1 < o0) { const double q = o0; o0 = o1; o1 = q; const int w = v0; v0 = v1; v1 = w; }
        if (o2 < o1) {
            { const double q = o1; o1 = o2; o2 = q; const int w = v1; v1 = v2; v2 = w; }
            if (o1 < o0) { const double q = o0; o0 = o1; o1 = q; const int w = v0; v0 = v1; v1 = w; }
        }
        tri.verts = (uint32_t)v0 | ((uint32_t)v1 << 8) | ((uint32_t)v2 << 16);
        tri.bin = ok ? (uint32_t)tri_bin(tri.mid) : 0u;
    }
    // slots: one global atomic per BLOCK (a same-address atomic with return costs ~12 ns; 3375 per-wave ones were 40 us)
    __shared__ unsigned int wave_cnt[kTriBlock / 64], block_base;
    const unsigned long long m = __ballot(ok);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) wave_cnt[wv] = (unsigned int)__builtin_popcountll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int tot = 0;
        for (int w = 0; w < kTriBlock / 64; ++w) tot += wave_cnt[w];
        block_base = tot ? atomicAdd(count, tot) : 0u;
    }
    __syncthreads();
    if (ok) {
        unsigned int base = block_base;
        for (int w = 0; w < wv; ++w) base += wave_cnt[w];
        out[base + (unsigned int)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = tri;
        if (bin_hist) atomicAdd(&lhist[tri.bin], 1u);
    }
    if (bin_hist) {
        __syncthreads();
        for (int b = threadIdx.x; b < kTriBins; b += kTriBlock)
            if (lhist[b]) atomicAdd(&bin_hist[b], lhist[b]);
    }
}

// exclusive scan of the 4096 bucket counts (one 256-thread block, 16 buckets per thread: a 1024-thread workgroup waits for sixteen
// free wave slots on one CU -- inside a batch of 171-Mpixel frames this 3 us kernel took 270 - 400 us); sets the scatter cursors and
// re-zeroes the histogram for the next table
constexpr int kScanThreads = 256, kScanPer = kTriBins / kScanThreads;
__device__ __forceinline__ void tri_bin_scan_body(unsigned int *__restrict__ hist, unsigned int *__restrict__ off /* kTriBins + 1 */,
                                                            unsigned int *__restrict__ cursor) {
    __shared__ unsigned int wave_tot[kScanThreads / 64];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    unsigned int h[kScanPer], s = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; j += 4) {
        const uint4 q = *reinterpret_cast<const uint4 *>(hist + kScanPer * t + j);
        h[j] = q.x, h[j + 1] = q.y, h[j + 2] = q.z, h[j + 3] = q.w;
        s += q.x + q.y + q.z + q.w;
    }
    unsigned int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    unsigned int base = 0;
    for (int i = 0; i < wv; ++i) base += wave_tot[i];
    unsigned int e = base + inc - s;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) {
        off[kScanPer * t + j] = e;
        cursor[kScanPer * t + j] = e;
        hist[kScanPer * t + j] = 0;
        e += h[j];
    }
    if (t == kScanThreads - 1) off[kTriBins] = e;
}

// Bucket scatter.  A block ranks its 1024 triangles inside their buckets with LDS atomics and reserves each touched
// bucket's range with ONE global atomic (per-triangle global cursor atomics-with-return serialised on the dense buckets:
// 845 triangles in the fullest one, ~12 ns each, 41 us for the kernel).
__device__ __forceinline__ void tri_scatter_body(const DTri *__restrict__ in, const unsigned int *__restrict__ n_p, unsigned int *cursor,
                                                           DTri *__restrict__ sorted) {
    __shared__ unsigned int cnt[kTriBins];  // per-bucket count of this block, then the bucket's reserved base
    const unsigned int n = *n_p, i = blockIdx.x * kTriBlock + threadIdx.x;
    if (blockIdx.x * kTriBlock >= n) return;
    for (int b = threadIdx.x; b < kTriBins; b += kTriBlock) cnt[b] = 0;
    __syncthreads();
    DTri t = {0.0, 0.0, 0u, 0u};
    unsigned int rank = 0;
    if (i < n) {
        t = in[i];
        rank = atomicAdd(&cnt[t.bin], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kTriBins; b += kTriBlock) {
        const unsigned int c = cnt[b];
        if (c) cnt[b] = atomicAdd(&cursor[b], c);
    }
    __syncthreads();
    if (i < n) sorted[cnt[t.bin] + rank] = t;  // order inside a bucket is irrelevant: votes are counts
}

// Orders every bucket of a scattered table by ratio_long (one 256-thread workgroup per bucket, bitonic network on
// (ratio_long, slot) pairs in LDS).  Only the REFERENCE table needs it: a wave of the vote kernel owns 64 consecutive ref
// triangles, and the narrower their ratio_long window, the fewer candidates it has to test.  Buckets of more than 1024
// triangles (none in practice: the fullest holds ~850) stay unordered -- slower, not wrong.  This replaced a host std::sort of
// the 34 220 triangles between two blocking copies: ~4 ms during which every worker that had finished its first detection
// waited for the reference table.
constexpr int kBucketSortCap = 1024;
__global__ __launch_bounds__(256) void tri_bucket_sort_kernel(DTri *__restrict__ tris, const unsigned int *__restrict__ off) {
    __shared__ double key[kBucketSortCap];
    __shared__ unsigned short slot[kBucketSortCap];
    __shared__ DTri item[kBucketSortCap];
    const unsigned int b0 = off[blockIdx.x], n = off[blockIdx.x + 1] - b0;
    if (n < 2 || n > (unsigned int)kBucketSortCap) return;
    unsigned int np = 2;
    while (np < n) np <<= 1;
    for (unsigned int i = threadIdx.x; i < np; i += 256) {
        if (i < n) item[i] = tris[b0 + i];
        key[i] = i < n ? item[i].lng : __builtin_huge_val();  // ratios are finite: the pads sort last
        slot[i] = (unsigned short)i;
    }
    __syncthreads();
    for (unsigned int k = 2; k <= np; k <<= 1)
        for (unsigned int j = k >> 1; j >= 1; j >>= 1) {
            for (unsigned int i = threadIdx.x; i < np; i += 256) {
                const unsigned int p = i ^ j;
                if (p > i) {
                    const bool up = (i & k) == 0;
                    const double a = key[i], c = key[p];
                    if (up ? c < a : a < c) {
                        key[i] = c;
                        key[p] = a;
                        const unsigned short q = slot[i];
                        slot[i] = slot[p];
                        slot[p] = q;
                    }
                }
            }
            __syncthreads();
        }
    for (unsigned int i = threadIdx.x; i < n; i += 256) tris[b0 + i] = item[slot[i]];
}

// The tgt table is bucketed by ratio_mid (tri_scatter_kernel); the ref table is sorted by (ratio_mid bucket,
// ratio_long) once per batch on the host.  One wave owns 64 consecutive ref triangles: a narrow bucket range
// [bmin, bmax] and a narrow ratio_long window [lmin, lmax].  It streams the tgt triangles of buckets bmin-1 ..
// bmax+1 (a superset of every possible match), keeps those whose ratio_long can match anything in the window
// (wave-wide compaction into LDS), and every lane applies the reference's exact test (:335-339) to the kept ones.
// Votes collect in LDS; each wave then adds its non-zero entries to the global 64 x 64 matrix.
__device__ __forceinline__ double lane_f64(double x, int lane) {  // wave-uniform lane index
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Per group of 64 consecutive reference triangles: the ratio windows and the bucket range a candidate must fall into to match
// ANY triangle of the group.  The reference table is the same for every target of a batch, so these are computed once per batch
// (one wave per group) instead of by four 64-lane f64 shuffle reductions per work item in every vote kernel -- 48 dependent
// ds_bpermute round trips per item in a kernel of 1024 lone waves that has nothing to hide them behind.
struct RefGroup {
    double lmin, lmax, mmin, mmax;
    int bmin, bmax;
};
__global__ __launch_bounds__(64) void tri_group_windows_kernel(const DTri *__restrict__ rt_sorted, const unsigned int *__restrict__ nr_p,
                                                               RefGroup *__restrict__ out) {
    const unsigned int nr = *nr_p, first = blockIdx.x * 64u, lane = threadIdx.x;
    if (first >= nr) return;
    const DTri a = rt_sorted[first + lane < nr ? first + lane : nr - 1];  // tail lanes replicate the last triangle
    double lmin = a.lng, lmax = a.lng, mmin = a.mid, mmax = a.mid;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lmin = fmin(lmin, __shfl_xor(lmin, off, 64));
        lmax = fmax(lmax, __shfl_xor(lmax, off, 64));
        mmin = fmin(mmin, __shfl_xor(mmin, off, 64));
        mmax = fmax(mmax, __shfl_xor(mmax, off, 64));
    }
    if (lane == 0) {
        RefGroup g;
        g.lmin = lmin;
        g.lmax = lmax;
        g.mmin = mmin;
        g.mmax = mmax;
        g.bmin = tri_bin(rt_sorted[first].mid);
        g.bmax = tri_bin(rt_sorted[min(first + 63u, nr - 1u)].mid);
        out[blockIdx.x] = g;
    }
}

// slices of a ref group's candidate range (dense buckets would otherwise leave a few very long waves) x persistent one-wave
// blocks, each flushing its LDS votes once.  Measured (blocks, slices) -> us per frame: (1024, 8) 62, (2048, 8) 74,
// (2048, 16) 69, (1024, 4) 79, (4096, 8) 79: more blocks pay for more flushes, fewer slices for imbalance.
constexpr int kVoteSlices = 8;
constexpr int kVoteBlocks = 1024;
// The blocks flush their LDS votes into one of `copies` copies of the matrix (block b -> copy b % copies) and the host adds the
// copies up: same-address device-scope atomics retire one per 15 .. 50 ns (stats_resident.hpp), and the pairs that collect votes
// collect them from hundreds of blocks.  Measured (AB_VOTE_COPIES = 1 / 4 / 16): kernel 83 / 80 / 80 us on average inside the batch
// (56.7 / 56.0 alone), stage median 16.3-16.6 / 15.8-15.9 / 15.7-16.3 ms: the flush is a small part of the kernel, 4 copies kept.
constexpr int kVoteCopiesMax = 16;
int vote_copies() {
    static const int c = [] {
        const char *e = ab_dev_env("AB_VOTE_COPIES");
        const int v = e ? atoi(e) : 4;
        return v < 1 ? 1 : (v > kVoteCopiesMax ? kVoteCopiesMax : v);
    }();
    return c;
}
// one work item: slice `item % kVoteSlices` of the candidates of reference group `item / kVoteSlices`, by one wave, into the LDS matrix
constexpr int kVoteLdsStride = kVoteDim + 1;  // LDS rows are 65 words apart: for one candidate every voting lane targets the SAME column, and
                                              // with a stride of 64 (a multiple of the bank count) all of those atomics would land in one bank
__device__ __forceinline__ void tri_vote_item(unsigned int item, int lane, unsigned int nr, const DTri *__restrict__ rt_sorted, const DTri *__restrict__ tt_sorted,
                                              const unsigned int *__restrict__ bin_off, const RefGroup *__restrict__ groups, unsigned int *votes,
                                              const unsigned int kVoteSlices = ::kVoteSlices) {
    const unsigned int first = (item / kVoteSlices) * 64, slice = item % kVoteSlices, r = first + lane;
    const bool have = r < nr;
    const DTri a = rt_sorted[have ? r : nr - 1];  // tail lanes replicate the last triangle (they never vote)
    const RefGroup grp = groups[item / kVoteSlices];  // (uniform: scalar loads)
    // a candidate can match SOME triangle of the group only inside the group's ratio windows (widened by the tolerance)
    const double win_lo = grp.lmin - kTriangleTolerance * 1.0001, win_hi = grp.lmax + kTriangleTolerance * 1.0001;
    const double mwin_lo = grp.mmin - kTriangleTolerance * 1.0001, mwin_hi = grp.mmax + kTriangleTolerance * 1.0001;
    const int bmin = grp.bmin, bmax = grp.bmax;
    unsigned int q0 = bin_off[bmin > 0 ? bmin - 1 : 0], q1 = bin_off[(bmax < kTriBins - 1 ? bmax + 1 : kTriBins - 1) + 1];
    {
        const unsigned int per = (q1 - q0 + kVoteSlices - 1) / kVoteSlices;
        q0 = min(q0 + slice * per, q1);
        q1 = min(q0 + per, q1);
    }
    // nothing else hides the table's load latency, so the next 64 candidates are fetched while the current ones are tested
    DTri nxt = {0.0, 0.0, 0u, 0u};
    if (q0 + lane < q1) nxt = tt_sorted[q0 + lane];
    for (unsigned int base = q0; base < q1; base += 64) {
        const unsigned int idx = base + lane;
        const DTri t = nxt;
        if (idx + 64 < q1) nxt = tt_sorted[idx + 64];
        const bool keep = idx < q1 && t.lng >= win_lo && t.lng <= win_hi && t.mid >= mwin_lo && t.mid <= mwin_hi;
        // the kept candidates stay in their lanes' registers and are broadcast one by one through the scalar unit
        // (v_readlane): staging them in LDS cost two dependent LDS round trips per candidate with nothing to hide them
        unsigned long long m = __ballot(keep);
        while (m) {
            const int q = (int)__builtin_ctzll(m);
            m &= m - 1;
            const double c_mid = lane_f64(t.mid, q), c_lng = lane_f64(t.lng, q);
            const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)t.verts, q);
            if (!have || fabs(a.mid - c_mid) > kTriangleTolerance || fabs(a.lng - c_lng) > kTriangleTolerance) continue;
#pragma unroll
            for (int p = 0; p < 3; ++p) atomicAdd(&votes[((a.verts >> (8 * p)) & 255u) * kVoteLdsStride + ((b >> (8 * p)) & 255u)], 1u);
        }
    }
}

// round 2 .. 5a: 1024 persistent one-wave blocks, each walking ~4 items and flushing its own LDS matrix (AB_VOTE_WAVES=1 keeps it)
__device__ __forceinline__ void tri_vote_body(const DTri *__restrict__ rt_sorted, const unsigned int *__restrict__ nr_p,
                                                      const DTri *__restrict__ tt_sorted, const unsigned int *__restrict__ bin_off,
                                                      unsigned int *__restrict__ votes_out /* copies x 64 x 64, zeroed */, int copies,
                                                      const RefGroup *__restrict__ groups, unsigned int *__restrict__ tgt_count_to_clear) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *tgt_count_to_clear = 0;  // tri_scatter_kernel was its last reader
    __shared__ unsigned int votes[kVoteDim * kVoteLdsStride];
    const int lane = threadIdx.x;
    const unsigned int nr = *nr_p, items = ((nr + 63u) / 64u) * kVoteSlices;
    if (blockIdx.x >= items) return;
    for (int i = lane; i < kVoteDim * kVoteLdsStride; i += 64) votes[i] = 0;
    __syncthreads();
    for (unsigned int item = blockIdx.x; item < items; item += gridDim.x) tri_vote_item(item, lane, nr, rt_sorted, tt_sorted, bin_off, groups, votes);
    __syncthreads();
    unsigned int *mine = votes_out + (size_t)(blockIdx.x % (unsigned int)copies) * (kVoteDim * kVoteDim);
    for (int row = 0; row < kVoteDim; ++row) {  // lane = column
        const unsigned int v = votes[row * kVoteLdsStride + lane];
        if (v) atomicAdd(&mine[row * kVoteDim + lane], v);
    }
}

// Round 5: ONE item per wave, kVoteWaves waves per block sharing the block's LDS matrix.  An item is a chain of three dependent
// global round trips (the group's windows -> the bucket offsets -> the first candidates) in front of a few dozen pair tests; a
// persistent wave walked four such chains one after the other with nothing beside it on its SIMD to hide them (90 us per group of
// four frames, 55 of them whatever the frame count).  Now every chain of a frame is in flight at once, and a block's clear + flush
// of the 64 x 64 matrix is shared by its waves (one-item one-wave blocks were tried in round 4: the flush per item made them slower).
constexpr int kVoteWaves = 8;
constexpr int kVoteItemsMax = ((kMaxTris + 63) / 64) * kVoteSlices;
__device__ __forceinline__ void tri_vote_wide_body(const DTri *__restrict__ rt_sorted, const unsigned int *__restrict__ nr_p, const DTri *__restrict__ tt_sorted,
                                                   const unsigned int *__restrict__ bin_off, unsigned int *__restrict__ votes_out, int copies,
                                                   const RefGroup *__restrict__ groups, unsigned int *__restrict__ tgt_count_to_clear, unsigned int slices) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *tgt_count_to_clear = 0;  // tri_scatter_kernel was its last reader
    __shared__ unsigned int votes[kVoteDim * kVoteLdsStride];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned int nr = *nr_p, items = ((nr + 63u) / 64u) * slices;
    if (blockIdx.x * kVoteWaves >= items) return;  // (block-uniform)
    for (int i = threadIdx.x; i < kVoteDim * kVoteLdsStride; i += kVoteWaves * 64) votes[i] = 0;
    __syncthreads();
    const unsigned int item = blockIdx.x * kVoteWaves + wv;  // consecutive items: slices of one reference group
    if (item < items) tri_vote_item(item, lane, nr, rt_sorted, tt_sorted, bin_off, groups, votes, slices);
    __syncthreads();
    unsigned int *mine = votes_out + (size_t)(blockIdx.x % (unsigned int)copies) * (kVoteDim * kVoteDim);
    for (int row = wv; row < kVoteDim; row += kVoteWaves) {  // lane = column
        const unsigned int v = votes[row * kVoteLdsStride + lane];
        if (v) atomicAdd(&mine[row * kVoteDim + lane], v);
    }
}

// ---- the kernels above as launches: one table, or the target tables of a GROUP of frames (blockIdx.y = frame; see detect.hip) ---
constexpr int kTriGroupMax = 8;
struct TriGroup {
    int n;
    int limit[kTriGroupMax];
    DTri *raw[kTriGroupMax], *sorted[kTriGroupMax];
    unsigned int *count[kTriGroupMax], *bin_hist[kTriGroupMax], *bin_off[kTriGroupMax], *cursor[kTriGroupMax], *votes[kTriGroupMax];
};
__global__ __launch_bounds__(kTriBlock) void tri_build_kernel(const StarXY stars, int limit, DTri *__restrict__ out, unsigned int *count,
                                                         unsigned int *__restrict__ bin_hist, unsigned int *__restrict__ votes_to_clear, int vote_words) { AB_LATENCY_KERNEL_PRIO();
    tri_build_body(stars.xy, limit, out, count, bin_hist, votes_to_clear, vote_words);
}
__global__ __launch_bounds__(kTriBlock) void tri_build_many_kernel(const TriGroup g, const StarXY *__restrict__ stars, int vote_words) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y;
    tri_build_body(stars[f].xy, g.limit[f], g.raw[f], g.count[f], g.bin_hist[f], g.votes[f], vote_words);
}
__global__ __launch_bounds__(kScanThreads) void tri_bin_scan_kernel(unsigned int *__restrict__ hist, unsigned int *__restrict__ off, unsigned int *__restrict__ cursor) { AB_LATENCY_KERNEL_PRIO();
    tri_bin_scan_body(hist, off, cursor);
}
__global__ __launch_bounds__(kScanThreads) void tri_bin_scan_many_kernel(const TriGroup g) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y;
    tri_bin_scan_body(g.bin_hist[f], g.bin_off[f], g.cursor[f]);
}
__global__ __launch_bounds__(kTriBlock) void tri_scatter_kernel(const DTri *__restrict__ in, const unsigned int *__restrict__ n_p, unsigned int *cursor,
                                                           DTri *__restrict__ sorted) { AB_LATENCY_KERNEL_PRIO();
    tri_scatter_body(in, n_p, cursor, sorted);
}
__global__ __launch_bounds__(kTriBlock) void tri_scatter_many_kernel(const TriGroup g) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y;
    tri_scatter_body(g.raw[f], g.count[f], g.cursor[f], g.sorted[f]);
}
__global__ __launch_bounds__(64) void tri_vote_kernel(const DTri *__restrict__ rt_sorted, const unsigned int *__restrict__ nr_p, const DTri *__restrict__ tt_sorted,
                                                      const unsigned int *__restrict__ bin_off, unsigned int *__restrict__ votes_out, int copies,
                                                      const RefGroup *__restrict__ groups, unsigned int *__restrict__ tgt_count_to_clear) { AB_LATENCY_KERNEL_PRIO();
    tri_vote_body(rt_sorted, nr_p, tt_sorted, bin_off, votes_out, copies, groups, tgt_count_to_clear);
}
__global__ __launch_bounds__(64) void tri_vote_many_kernel(const TriGroup g, const DTri *__restrict__ rt_sorted, const unsigned int *__restrict__ nr_p, int copies,
                                                           const RefGroup *__restrict__ groups) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y;
    tri_vote_body(rt_sorted, nr_p, g.sorted[f], g.bin_off[f], g.votes[f], copies, groups, g.count[f]);
}
__global__ __launch_bounds__(kVoteWaves * 64) void tri_vote_wide_kernel(const DTri *__restrict__ rt_sorted, const unsigned int *__restrict__ nr_p, const DTri *__restrict__ tt_sorted,
                                                                        const unsigned int *__restrict__ bin_off, unsigned int *__restrict__ votes_out, int copies,
                                                                        const RefGroup *__restrict__ groups, unsigned int *__restrict__ tgt_count_to_clear, unsigned int slices) { AB_LATENCY_KERNEL_PRIO();
    tri_vote_wide_body(rt_sorted, nr_p, tt_sorted, bin_off, votes_out, copies, groups, tgt_count_to_clear, slices);
}
__global__ __launch_bounds__(kVoteWaves * 64) void tri_vote_wide_many_kernel(const TriGroup g, const DTri *__restrict__ rt_sorted, const unsigned int *__restrict__ nr_p, int copies,
                                                                             const RefGroup *__restrict__ groups, unsigned int slices) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y;
    tri_vote_wide_body(rt_sorted, nr_p, g.sorted[f], g.bin_off[f], g.votes[f], copies, groups, g.count[f], slices);
}
unsigned int vote_wide_slices() {  // slices of a reference group's candidate range in the wide form (AB_VOTE_SLICES)
    static const unsigned int v = [] {
        const char *e = ab_dev_env("AB_VOTE_SLICES");
        const int x = e ? atoi(e) : 8;  // (8 / 4 / 2 / 1 measured: 9.4 / 9.6 / 9.6 / 10.0 ms for the stage)
        return (unsigned int)(x < 1 ? 1 : (x > 8 ? 8 : x));
    }();
    return v;
}
bool vote_wide() {
    static const bool on = [] {
        const char *e = ab_dev_env("AB_VOTE_WAVES");
        return !(e && atoi(e) == 1);
    }();
    return on;
}

// the `copies` partial vote matrices of every frame of a group added up, straight into pinned host memory (round 4: the group's
// 16 x 16 KB slots used to come back as one 1 MB blit and the host added them: four launches + this one, no copy out)
__global__ __launch_bounds__(256) void votes_reduce_many_kernel(const TriGroup g, int copies, uint32_t *__restrict__ out) { AB_LATENCY_KERNEL_PRIO();
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kVoteDim * kVoteDim) return;
    uint32_t s = 0;
    for (int c = 0; c < copies; ++c) s += g.votes[f][(size_t)c * kVoteDim * kVoteDim + i];
    out[(size_t)f * kVoteDim * kVoteDim + i] = s;
}

// device-side layout of the matcher's workspace (AB_WS_REGISTER)
struct MatchWs {
    DTri *ref_tris, *ref_sorted, *tgt_tris, *tgt_sorted;
    unsigned int *counts;  // [0] ref, [1] tgt
    unsigned int *bin_hist, *bin_off, *cursor;
    unsigned int *votes;
    RefGroup *groups;  // per 64 reference triangles (tri_group_windows_kernel)
};

int match_ws(ab_ctx *ctx, MatchWs *w) {
    const size_t tri_bytes = (size_t)kMaxTris * sizeof(DTri), vote_bytes = (size_t)kVoteCopiesMax * kVoteDim * kVoteDim * sizeof(unsigned int);
    const size_t bin_words = 64 + 3 * (size_t)kTriBins + 64;
    const size_t group_bytes = (size_t)((kMaxTris + 63) / 64) * sizeof(RefGroup);
    const size_t total = 4 * tri_bytes + bin_words * sizeof(unsigned int) + vote_bytes + group_bytes;
    char *p = nullptr;
    const void *before = ctx->ws[AB_WS_REGISTER];
    AB_TRY(ab_workspace(ctx, AB_WS_REGISTER, total, (void **)&p));
    if (p != before)  // bin_hist is kept zero between uses (tri_bin_scan_kernel clears what it read): only a fresh workspace needs it
        AB_HIP(ctx, hipMemsetAsync(p + 4 * tri_bytes, 0, bin_words * sizeof(unsigned int), ctx->stream));
    w->ref_tris = (DTri *)p;
    w->tgt_tris = (DTri *)(p + tri_bytes);
    w->tgt_sorted = (DTri *)(p + 2 * tri_bytes);
    w->ref_sorted = (DTri *)(p + 3 * tri_bytes);
    unsigned int *u = (unsigned int *)(p + 4 * tri_bytes);
    w->counts = u;
    w->bin_hist = u + 64;
    w->bin_off = w->bin_hist + kTriBins;  // kTriBins + 1 entries
    w->cursor = w->bin_off + kTriBins + 32;
    w->votes = u + bin_words;
    w->groups = (RefGroup *)((char *)w->votes + vote_bytes);
    return AB_OK;
}

// upload the first <= 60 stars and build their triangle table (which = 0 ref, 1 tgt), bucketed by ratio_mid; the ref table's
// buckets are ordered by ratio_long as well.  Both tables use w's bucket arrays: the ref table is built once per batch on the
// caller's context, whose own tgt slots are never used.
int gpu_build_triangles(ab_ctx *ctx, const MatchWs &w, const std::vector<Pt> &stars, int which) {
    const int limit = (int)std::min<size_t>(stars.size(), kTriLimit);
    StarXY xy;  // 960 B of kernel arguments: no staging copy, nothing to keep alive
    memset(&xy, 0, sizeof xy);
    for (int i = 0; i < limit; ++i) {
        xy.xy[2 * i] = stars[i][0];
        xy.xy[2 * i + 1] = stars[i][1];
    }
    DTri *raw = which ? w.tgt_tris : w.ref_tris, *sorted = which ? w.tgt_sorted : w.ref_sorted;
    // the target table's counter is left at zero by the previous frame's vote kernel (and by the fresh workspace's memset); the
    // reference table is built once per batch
    if (!which) AB_HIP(ctx, hipMemsetAsync(w.counts, 0, sizeof(unsigned int), ctx->stream));
    const int total = limit * limit * limit;
    const int vote_words = vote_copies() * kVoteDim * kVoteDim;
    const bool clears = which && limit >= 3 && (total + kTriBlock - 1) / kTriBlock * kTriBlock >= vote_words;  // enough threads to clear the votes
    if (limit >= 3)
        hipLaunchKernelGGL(tri_build_kernel, dim3((total + kTriBlock - 1) / kTriBlock), dim3(kTriBlock), 0, ctx->stream, xy, limit, raw, w.counts + which, w.bin_hist,
                           clears ? w.votes : (unsigned int *)nullptr, vote_words);
    if (which && !clears) AB_HIP(ctx, hipMemsetAsync(w.votes, 0, (size_t)vote_words * sizeof(unsigned int), ctx->stream));
    // (the scan leaves bin_hist zeroed again)
    hipLaunchKernelGGL(tri_bin_scan_kernel, dim3(1), dim3(kScanThreads), 0, ctx->stream, w.bin_hist, w.bin_off, w.cursor);
    hipLaunchKernelGGL(tri_scatter_kernel, dim3((kMaxTris + kTriBlock - 1) / kTriBlock), dim3(kTriBlock), 0, ctx->stream, raw, w.counts + which, w.cursor, sorted);
    if (!which) {
        hipLaunchKernelGGL(tri_bucket_sort_kernel, dim3(kTriBins), dim3(256), 0, ctx->stream, sorted, w.bin_off);
        hipLaunchKernelGGL(tri_group_windows_kernel, dim3((kMaxTris + 63) / 64), dim3(64), 0, ctx->stream, (const DTri *)sorted, (const unsigned int *)w.counts,
                           w.groups);
    }
    AB_HIP(ctx, hipGetLastError());
    return AB_OK;
}

// votes of the current ref / tgt triangle tables -> host (kVoteDim x kVoteDim)
int gpu_votes(ab_ctx *ctx, const MatchWs &w, const unsigned int *ref_count, std::vector<uint32_t> *votes) {
    // (w.votes was cleared by this frame's tri_build_kernel or the memset beside it; the kernel resets the target table's counter
    // for the next frame)
    const int copies = vote_copies();
    if (vote_wide())
        hipLaunchKernelGGL(tri_vote_wide_kernel, dim3((kVoteItemsMax + kVoteWaves - 1) / kVoteWaves), dim3(kVoteWaves * 64), 0, ctx->stream, w.ref_sorted, ref_count, w.tgt_sorted,
                           w.bin_off, w.votes, copies, (const RefGroup *)w.groups, w.counts + 1, vote_wide_slices());
    else
        hipLaunchKernelGGL(tri_vote_kernel, dim3(kVoteBlocks), dim3(64), 0, ctx->stream, w.ref_sorted, ref_count, w.tgt_sorted, w.bin_off, w.votes, copies,
                           (const RefGroup *)w.groups, w.counts + 1);
    AB_HIP(ctx, hipGetLastError());
    const size_t words = (size_t)copies * kVoteDim * kVoteDim;
    void *pin = nullptr;  // (read back into pinned memory: a copy into a std::vector is staged and synchronised by the runtime)
    AB_TRY(ab_pinned(ctx, words * sizeof(uint32_t), &pin));
    AB_HIP(ctx, hipMemcpyAsync(pin, w.votes, words * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t *h = (const uint32_t *)pin;
    votes->assign(h, h + kVoteDim * kVoteDim);
    for (int c = 1; c < copies; ++c)
        for (int i = 0; i < kVoteDim * kVoteDim; ++i) (*votes)[i] += h[(size_t)c * kVoteDim * kVoteDim + i];
    return AB_OK;
}

// The target-side matcher state of a group of frames: G x {raw table, bucketed table, counter, bucket arrays, vote matrices}, the
// frames' star coordinates, kept in AB_WS_REGISTER_GROUP.  Zero where the kernels expect zero (counters, bucket histograms) when the
// workspace is new; they leave it that way.
struct MatchGroupWs {
    TriGroup g;
    StarXY *stars;
    unsigned int *votes_all;  // G x copies x 64 x 64, contiguous: one copy brings all of them back
};
int match_group_ws(ab_ctx *ctx, int G, MatchGroupWs *w) {
    const size_t tri_bytes = (size_t)kMaxTris * sizeof(DTri), vote_bytes = (size_t)kVoteCopiesMax * kVoteDim * kVoteDim * sizeof(unsigned int);
    const size_t word_block = (64 + 3 * (size_t)kTriBins + 64) * sizeof(unsigned int);
    const size_t per = 2 * tri_bytes + word_block, stars_bytes = (size_t)kTriGroupMax * sizeof(StarXY);
    const size_t total = (size_t)kTriGroupMax * (per + vote_bytes) + stars_bytes;
    char *p = nullptr;
    const void *before = ctx->ws[AB_WS_REGISTER_GROUP];
    AB_TRY(ab_workspace(ctx, AB_WS_REGISTER_GROUP, total, (void **)&p));
    if (p != before) AB_HIP(ctx, hipMemsetAsync(p, 0, total, ctx->stream));
    memset(&w->g, 0, sizeof w->g);
    w->g.n = G;
    w->votes_all = (unsigned int *)(p + (size_t)kTriGroupMax * per);
    w->stars = (StarXY *)(p + (size_t)kTriGroupMax * (per + vote_bytes));
    for (int f = 0; f < kTriGroupMax; ++f) {
        char *q = p + (size_t)f * per;
        w->g.raw[f] = (DTri *)q;
        w->g.sorted[f] = (DTri *)(q + tri_bytes);
        unsigned int *u = (unsigned int *)(q + 2 * tri_bytes);
        w->g.count[f] = u;
        w->g.bin_hist[f] = u + 64;
        w->g.bin_off[f] = u + 64 + kTriBins;
        w->g.cursor[f] = u + 64 + 2 * kTriBins + 32;
        w->g.votes[f] = w->votes_all + (size_t)f * (vote_bytes / sizeof(unsigned int));
    }
    return AB_OK;
}

// the vote matrices of G targets' star lists against the reference table of `ref_ws`: five launches, one copy in, the result written
// straight into pinned memory
int gpu_match_group(ab_ctx *ctx, const MatchWs &ref_ws, const std::vector<Pt> *stars /* [G] */, int G, std::vector<uint32_t> *votes /* [G] */) {
    MatchGroupWs w;
    AB_TRY(match_group_ws(ctx, G, &w));
    const int copies = vote_copies(), vote_words = copies * kVoteDim * kVoteDim;
    const size_t per_frame_words = (size_t)kVoteDim * kVoteDim;
    void *pin = nullptr;
    const size_t stars_bytes = (size_t)G * sizeof(StarXY), votes_bytes = (size_t)G * per_frame_words * sizeof(uint32_t);
    AB_TRY(ab_pinned(ctx, stars_bytes + votes_bytes, &pin));
    StarXY *hs = (StarXY *)pin;
    memset(hs, 0, stars_bytes);
    int max_limit = 0;
    for (int f = 0; f < G; ++f) {
        const int limit = (int)std::min<size_t>(stars[f].size(), kTriLimit);
        w.g.limit[f] = limit;
        max_limit = std::max(max_limit, limit);
        for (int i = 0; i < limit; ++i) {
            hs[f].xy[2 * i] = stars[f][i][0];
            hs[f].xy[2 * i + 1] = stars[f][i][1];
        }
    }
    for (int f = 0; f < G; ++f) votes[f].assign((size_t)kVoteDim * kVoteDim, 0u);
    if (max_limit < 3) return AB_OK;
    AB_HIP(ctx, hipMemcpyAsync(w.stars, hs, stars_bytes, hipMemcpyHostToDevice, ctx->stream));
    // every table's builder clears that frame's vote matrices (its grid must have the threads: 1024-thread blocks over limit^3 >= 27)
    const int total = max_limit * max_limit * max_limit, blocks = std::max((total + kTriBlock - 1) / kTriBlock, (vote_words + kTriBlock - 1) / kTriBlock);
    hipLaunchKernelGGL(tri_build_many_kernel, dim3(blocks, G), dim3(kTriBlock), 0, ctx->stream, w.g, (const StarXY *)w.stars, vote_words);
    hipLaunchKernelGGL(tri_bin_scan_many_kernel, dim3(1, G), dim3(kScanThreads), 0, ctx->stream, w.g);
    hipLaunchKernelGGL(tri_scatter_many_kernel, dim3((kMaxTris + kTriBlock - 1) / kTriBlock, G), dim3(kTriBlock), 0, ctx->stream, w.g);
    if (vote_wide())
        hipLaunchKernelGGL(tri_vote_wide_many_kernel, dim3((kVoteItemsMax + kVoteWaves - 1) / kVoteWaves, G), dim3(kVoteWaves * 64), 0, ctx->stream, w.g,
                           (const DTri *)ref_ws.ref_sorted, (const unsigned int *)ref_ws.counts, copies, (const RefGroup *)ref_ws.groups, vote_wide_slices());
    else
        hipLaunchKernelGGL(tri_vote_many_kernel, dim3(kVoteBlocks, G), dim3(64), 0, ctx->stream, w.g, (const DTri *)ref_ws.ref_sorted, (const unsigned int *)ref_ws.counts,
                           copies, (const RefGroup *)ref_ws.groups);
    uint32_t *hv = (uint32_t *)((char *)pin + stars_bytes);
    hipLaunchKernelGGL(votes_reduce_many_kernel, dim3((kVoteDim * kVoteDim + 255) / 256, G), dim3(256), 0, ctx->stream, w.g, copies, hv);
    AB_HIP(ctx, hipGetLastError());
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int f = 0; f < G; ++f) votes[f].assign(hv + (size_t)f * per_frame_words, hv + (size_t)(f + 1) * per_frame_words);
    return AB_OK;
}

// normalize_for_detection + detect_stars(3.5 sigma) + top_n_stars of one frame (:134-160)
int frame_stars(ab_ctx *ctx, const float *img, int64_t rows, int64_t cols, std::vector<Pt> *out, const ab_pixel_xf *xf = nullptr,
                const double *bg = nullptr, const ab_frame_cand *cand = nullptr /* the frame's candidate lists, if its tiles left any */) {
    std::vector<ab_detected_star> stars;
    double m, s;
    // the normalised frame is never materialised: detection applies the transform on load.  xf: the frame's transform if the
    // caller has it already (a batch takes all its frames' percentiles in one go); otherwise detection derives it first
    if (xf && bg) {
        // transform and background known (a batch's reference frame, or a target taken frame by frame): the grouped detection with
        // a group of one -- the brightest candidates are chosen on the device and only their boxes are walked (round 5; the full
        // list of C3's reference frame, ~100 000 components, cost 3.8 ms of moments at the head of every call).  Same stars:
        // tests/test_gpu_detect_affine.py holds the grouped form to the frame-by-frame one.
        const double b2[1][2] = {{bg[0], bg[1]}};
        AB_TRY(ab_detect_stars_group_device(ctx, &img, 1, rows, cols, kDetectionSigma, xf, b2, kMaxStars, &stars, cand));
    } else {
        AB_TRY(ab_detect_stars_device(ctx, img, rows, cols, cols, kDetectionSigma, &stars, &m, &s, xf ? *xf : ab_pixel_xf(), kMaxStars,
                                      /*normalize_first=*/xf == nullptr, xf ? bg : nullptr));  // top_n_stars (:272-277)
    }
    out->clear();
    for (const auto &st : stars) {
        if (out->size() >= kMaxStars) break;  // top_n_stars (:272-277): detections are already sorted by flux
        out->push_back({st.x, st.y});
    }
    return AB_OK;
}

}  // namespace

// The reference frame's star list and triangle table, prepared on the caller's context while the workers already run
// their targets' detection (the targets' detection does not depend on it): wait() blocks until it is there.
struct RefTable {
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
    int rc = AB_OK;
    bool ok = false;  // enough reference stars to match at all
    std::vector<Pt> stars;
    void publish(int rc_, bool ok_) {
        {
            std::lock_guard<std::mutex> g(m);
            rc = rc_;
            ok = ok_;
            done = true;
        }
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> g(m);
        cv.wait(g, [&] { return done; });
        return rc;
    }
};

// one target against the reference being prepared (stars + triangle table in rt / ref_ws); all device work on wc's stream
static int register_one(ab_ctx *wc, const MatchWs &ref_ws, RefTable &rt, const float *ref, const float *tgt, int64_t rows, int64_t cols,
                        int num_threads, ab_affine_align_result *out, const ab_pixel_xf *tgt_xf, const double *tgt_bg) {
    MatchWs w;
    AB_TRY(match_ws(wc, &w));
    std::vector<Pt> ts;
    bool found = false;
    ab_trace trace("register_one");
    AB_TRY(frame_stars(wc, tgt, rows, cols, &ts, tgt_xf, tgt_bg));
    trace.mark("frame_stars");
    if (rt.wait() != AB_OK) return ab_set_error(wc, rt.rc, "the reference frame's detection failed");
    const std::vector<Pt> &rs = rt.stars;
    if (rt.ok && ts.size() >= kMinMatchesRigid) {
        AB_TRY(gpu_build_triangles(wc, w, ts, 1));
        MatchWs mixed = w;  // tgt table and votes of this worker; ref table of the caller
        mixed.ref_sorted = ref_ws.ref_sorted;
        mixed.groups = ref_ws.groups;
        std::vector<uint32_t> votes;
        AB_TRY(gpu_votes(wc, mixed, ref_ws.counts, &votes));
        trace.mark("triangles+votes");
        // an empty triangle table on either side leaves the votes at zero -> no matches, as :166-168
        const auto matches = matches_from_votes(rs, ts, votes.data(), kVoteDim);
        trace.mark("matches");
        found = transform_from_matches(matches, rows, cols, num_threads, out);
        trace.mark("ransac+fit");
    }
    if (found) return AB_OK;
    // fallback_phase_correlation (:243-270) on the ORIGINAL planes
    double dx, dy, conf;
    AB_TRY(ab_phase_correlate_device(wc, ref, rows, cols, cols, tgt, rows, cols, cols, &dx, &dy, &conf));
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

// G targets against the reference in lockstep (round 4): one launch per step of the detection and of the matcher for all of them,
// the host geometry frame by frame in between.  out / xf are indexed by frame, bg by position in the group.
static int register_group(ab_ctx *wc, const MatchWs &ref_ws, RefTable &rt, const float *ref, const float *const *targets, const size_t *frames, int G,
                          int64_t rows, int64_t cols, int num_threads, ab_affine_align_result *out, const ab_pixel_xf *xfs, const double (*bg)[2],
                          const std::function<int(size_t)> &frame_done /* called with a frame's index as soon as out[frame] is final */,
                          const ab_frame_cand *cands = nullptr /* [G], by position in the group */) {
    const float *imgs[kTriGroupMax];
    ab_pixel_xf xf[kTriGroupMax];
    for (int k = 0; k < G; ++k) {
        imgs[k] = targets[frames[k]];
        xf[k] = xfs[frames[k]];
    }
    std::vector<ab_detected_star> det[kTriGroupMax];
    ab_upload_trace("group starts detection, first frame", (long)frames[0]);
    AB_TRY(ab_detect_stars_group_device(wc, imgs, G, rows, cols, kDetectionSigma, xf, bg, kMaxStars, det, cands));
    ab_upload_trace("group detected, first frame", (long)frames[0]);
    std::vector<Pt> ts[kTriGroupMax];
    for (int k = 0; k < G; ++k)
        for (const auto &st : det[k]) {
            if (ts[k].size() >= kMaxStars) break;  // top_n_stars (:272-277): detections are already sorted by flux
            ts[k].push_back({st.x, st.y});
        }
    if (rt.wait() != AB_OK) return ab_set_error(wc, rt.rc, "the reference frame's detection failed");
    const std::vector<Pt> &rs = rt.stars;
    bool found[kTriGroupMax] = {};
    if (rt.ok) {
        std::vector<Pt> live[kTriGroupMax];  // a frame with too few stars takes no part in the matching (:166-168): an empty list
        bool any = false;
        for (int k = 0; k < G; ++k)
            if (ts[k].size() >= kMinMatchesRigid) {
                live[k] = ts[k];
                any = true;
            }
        if (any) {
            std::vector<uint32_t> votes[kTriGroupMax];
            AB_TRY(gpu_match_group(wc, ref_ws, live, G, votes));
            ab_upload_trace("group votes back, first frame", (long)frames[0]);
            // (Round 6, measured and not kept: fitting the group's frames side by side on helper threads -- 0.6 -> 0.16 ms of host
            // arithmetic per group -- moved the step by 0.04 ms, inside the noise: the chains are not what the call waits for.)
            for (int k = 0; k < G; ++k) {
                if (live[k].empty()) continue;
                const auto matches = matches_from_votes(rs, ts[k], votes[k].data(), kVoteDim);
                found[k] = transform_from_matches(matches, rows, cols, num_threads, &out[frames[k]]);
                ab_upload_trace("frame fitted", (long)frames[k]);
                if (found[k]) AB_TRY(frame_done(frames[k]));  // (its warp starts while the next frame's RANSAC runs)
            }
        }
    }
    for (int k = 0; k < G; ++k) {
        if (found[k]) continue;
        // fallback_phase_correlation (:243-270) on the ORIGINAL planes
        ab_affine_align_result *o = &out[frames[k]];
        double dx, dy, conf;
        AB_TRY(ab_phase_correlate_device(wc, ref, rows, cols, cols, imgs[k], rows, cols, cols, &dx, &dy, &conf));
        memset(o, 0, sizeof *o);
        o->transform[0] = 1.0;
        o->transform[4] = 1.0;
        if (std::fabs(dx) > (double)cols * kMaxOffsetFraction || std::fabs(dy) > (double)rows * kMaxOffsetFraction || conf < 1.5) {
            o->method = kIdentity;
        } else {
            o->transform[2] = dx;
            o->transform[5] = dy;
            o->method = kPhaseCorr;
        }
        AB_TRY(frame_done(frames[k]));
    }
    return AB_OK;
}

// align_channel_affine (affine.rs:129-212) of n targets against ONE reference.  The reference's normalisation,
// detection and triangle table are computed once.  Targets are independent, so they are spread over up to
// ctx->register_workers host threads, each driving its own HIP stream and workspaces (child contexts cached in
// ctx): one frame's host geometry overlaps the other frames' GPU passes.  Results do not depend on the worker
// count (each out[i] equals a stand-alone align_channel_affine(reference, targets[i])).
// band_rows >= 0 (round 6, the row-band scheme): aligned[f] holds rows [band_row0, band_row0 + band_rows) of warp_image(target f, transform)
// only -- a rank warps ITS rows of a frame the moment the frame is fitted, overlapped with the other frames' estimates like the full warps.
int ab_register_frames_device(ab_ctx *ctx, const float *ref, const float *const *targets, size_t n, int64_t rows, int64_t cols, int num_threads,
                              ab_affine_align_result *out, float *const *aligned /* nullable: warp_image(target, transform) per target */,
                              const hipEvent_t *landed /* nullable: per target, the event after which its pixels are in HBM (an upload in flight) */,
                              int64_t band_row0, int64_t band_rows) {
    auto warp_out = [&](ab_ctx *wc, size_t f) -> int {
        if (band_rows >= 0) return ab_warp_rows_device(wc, targets[f], rows, cols, out[f].transform, rows, cols, band_row0, band_rows, aligned[f]);
        return ab_warp_device(wc, targets[f], rows, cols, out[f].transform, rows, cols, aligned[f]);
    };
    AB_HIP(ctx, hipSetDevice(ctx->device));
    if (ab_upload_trace_on() && !landed) ab_trace_t0() = std::chrono::steady_clock::now();  // (a host-fed call has set it)
    MatchWs w;
    AB_TRY(match_ws(ctx, &w));
    RefTable rt;
    // normalize_for_detection's percentiles of every frame of the batch (targets, then the reference) up front: two launches, 64
    // percentile workgroups side by side (see ab_normalize_params_many_device)
    std::vector<ab_pixel_xf> xfs;
    ab_bg_pipeline pipe;  // plane 0 = the reference, plane 1 + f = target f
    // whatever path leaves this function, tile launches still in flight on the auxiliary stream (they read the caller's frames)
    // are drained first
    struct AuxDrain {
        ab_ctx *c;
        ab_bg_pipeline *p;
        ~AuxDrain() {
            ab_bg_pipeline_end(p);
            if (p->on && c->pct_stream) (void)hipStreamSynchronize(c->pct_stream);
            if (p->on && c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
        }
    } aux_drain{ctx, &pipe};
    const bool have_xf = n >= 4 || landed != nullptr;
    // Round 4: frames that are still on the PCIe link (`landed`) go through the FED pipeline (ab_bg_pipeline_begin_fed): a chunk's
    // percentiles run on the device right before its tiles and wait for the chunk's upload events, so nothing is joined on the host
    // before the first tile launch.  Device-resident frames keep the percentiles of all frames up front (two launches + a join):
    // measured on the bench stack, same box, 12.99 ms for the stage against 13.34 fed (sixteen more launches in the tile stream's
    // way).  AB_PIPE_FED=1 feeds them too (the GPU tests hold the two orders to the same transforms).
    static const bool force_fed = ab_dev_env("AB_PIPE_FED") != nullptr;
    static const bool want_cand = ab_dev_env("AB_NO_CAND_LISTS") == nullptr;  // (developer A/B: the labelling reads the frames as in round 5)
    const bool fed = have_xf && (landed != nullptr || force_fed);
    if (fed) {
        static const int chunk = ab_dev_env("AB_TILE_CHUNK") ? std::max(atoi(ab_dev_env("AB_TILE_CHUNK")), 1) : 16;
        static const int fed_chunk = ab_dev_env("AB_FEED_CHUNK") ? std::max(atoi(ab_dev_env("AB_FEED_CHUNK")), 1) : 4;  // frames per launch while frames are still arriving
        std::vector<const float *> order;  // reference first
        std::vector<hipEvent_t> ev;
        order.push_back(ref);
        ev.push_back(nullptr);
        for (size_t i = 0; i < n; ++i) {
            order.push_back(targets[i]);
            ev.push_back(landed ? landed[i] : nullptr);
        }
        xfs.resize(n + 1);
        AB_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's frames are complete before any other stream reads them
        AB_TRY(ab_bg_pipeline_begin_fed(ctx, order.data(), n + 1, rows, cols, landed ? fed_chunk : chunk, landed ? ev.data() : nullptr, &pipe, want_cand));
        if (!pipe.on && landed)  // (planes too small for tiles, AB_TILE_LEGACY: the frames are waited for and the batch takes the path below)
            for (size_t i = 0; i < n; ++i)
                if (landed[i]) AB_HIP(ctx, hipEventSynchronize(landed[i]));
    }
    if (have_xf && !pipe.on) {
        std::vector<const float *> planes(targets, targets + n);
        planes.push_back(ref);
        xfs.resize(n + 1);
        AB_TRY(ab_normalize_params_many_device(ctx, planes.data(), n + 1, rows * cols, xfs.data()));
        ab_upload_trace("percentiles joined, planes", (long)(n + 1));
        // ... and their background tiles: the tile kernel runs on its own stream, a few frames per launch, the reference first,
        // while the workers already label the frames whose tiles are done (register_one blocks on its frame's chunk)
        static const int chunk = ab_dev_env("AB_TILE_CHUNK") ? atoi(ab_dev_env("AB_TILE_CHUNK")) : 16;  // (round 6: 16 frames per launch, 9.73 / 9.74 / 9.68 ms per step against 9.82 / 9.80 / 9.90 with 8: profiles/r06_priority_ab.txt)  // frames per launch (0: every frame's tiles in its own chain); measured 0 / 2 / 4 / 8 / 16 -> 17.6 / 17.5 / 17.2 / 17.0 / 17.4 ms for the stage
        if (chunk > 0) {
            std::vector<const float *> order;  // reference first
            std::vector<ab_pixel_xf> oxf;
            order.push_back(ref);
            oxf.push_back(xfs[n]);
            for (size_t i = 0; i < n; ++i) {
                order.push_back(targets[i]);
                oxf.push_back(xfs[i]);
            }
            AB_TRY(ab_bg_pipeline_begin(ctx, order.data(), n + 1, rows, cols, oxf.data(), chunk, &pipe, want_cand));
            ab_upload_trace("tile launches enqueued, planes", (long)(n + 1));
        }
    }
    auto prepare_reference = [&]() -> int {
        double bg[2];
        if (pipe.on) AB_TRY(ab_bg_pipeline_get(ctx, &pipe, 0, bg));
        if (pipe.xf_host) xfs[n] = pipe.xf_host[0];
        const ab_frame_cand ref_cand = ab_bg_pipeline_cand(&pipe, 0);
        AB_TRY(frame_stars(ctx, ref, rows, cols, &rt.stars, have_xf ? &xfs[n] : nullptr, pipe.on ? bg : nullptr, &ref_cand));
        if (rt.stars.size() < kMinMatchesRigid) return AB_OK;
        // reference table: built, bucketed by ratio_mid and ordered by ratio_long inside the buckets on the GPU
        AB_TRY(gpu_build_triangles(ctx, w, rt.stars, 0));
        AB_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the workers' vote kernels read it from their own streams
        return AB_OK;
    };
    hipStream_t warp_stream = nullptr;
    // the warp of a frame (f64 VALU) overlaps the other workers' latency-bound detection passes
    auto one = [&](ab_ctx *wc, size_t f) -> int {
        double bg[2];
        if (pipe.on) AB_TRY(ab_bg_pipeline_get(wc, &pipe, f + 1, bg));
        if (pipe.xf_host) xfs[f] = pipe.xf_host[f + 1];  // (each worker writes its own frames' entries)
        AB_TRY(register_one(wc, w, rt, ref, targets[f], rows, cols, num_threads, &out[f], have_xf ? &xfs[f] : nullptr, pipe.on ? bg : nullptr));
        if (aligned) {  // pair.rs:59-61
            // the warp needs nothing from the GPU but the frame: it goes to the batch's warp stream, so that this worker's
            // next frame does not queue up behind it on the worker's own stream
            hipStream_t keep = wc->stream;
            if (warp_stream) wc->stream = warp_stream;
            const int rc = warp_out(wc, f);
            wc->stream = keep;
            if (rc != AB_OK) return rc;
        }
        return AB_OK;
    };
    // Round 4: the targets go through detection and matching in GROUPS of kGroup (AB_REGISTER_GROUP, default 4; 1 = frame by frame as
    // in rounds 2 / 3): the stage was bound by the number of launches the four hardware queues serialise, not by their work.
    static const int kGroup = [] {
        const char *e = ab_dev_env("AB_REGISTER_GROUP");
        const int v = e ? atoi(e) : 4;
        return v < 1 ? 1 : (v > kTriGroupMax ? kTriGroupMax : v);
    }();
    const bool grouped = kGroup > 1 && pipe.on && have_xf;
    const size_t n_groups = grouped ? (n + (size_t)kGroup - 1) / (size_t)kGroup : 0;
    auto one_group = [&](ab_ctx *wc, size_t gi) -> int {
        size_t frames[kTriGroupMax];
        double bgs[kTriGroupMax][2];
        ab_frame_cand cands[kTriGroupMax];
        const size_t f0 = gi * (size_t)kGroup;
        const int G = (int)std::min<size_t>((size_t)kGroup, n - f0);
        for (int k = 0; k < G; ++k) {
            frames[k] = f0 + (size_t)k;
            AB_TRY(ab_bg_pipeline_get(wc, &pipe, frames[k] + 1, bgs[k]));
            cands[k] = ab_bg_pipeline_cand(&pipe, frames[k] + 1);  // (complete: the plane's chunk has run)
            if (pipe.xf_host) xfs[frames[k]] = pipe.xf_host[frames[k] + 1];  // (each worker writes its own frames' entries)
        }
        const std::function<int(size_t)> warp_frame = [&](size_t f) -> int {
            if (!aligned) return AB_OK;  // pair.rs:59-61
            hipStream_t keep = wc->stream;
            if (warp_stream) wc->stream = warp_stream;
            const int rc = warp_out(wc, f);
            wc->stream = keep;
            return rc;
        };
        return register_group(wc, w, rt, ref, targets, frames, G, rows, cols, num_threads, out, xfs.data(), bgs, warp_frame, cands);
    };
    const size_t n_jobs = grouped ? n_groups : n;
    const std::function<int(ab_ctx *, size_t)> job = grouped ? std::function<int(ab_ctx *, size_t)>(one_group) : std::function<int(ab_ctx *, size_t)>(one);
    const bool inline_run = std::min<size_t>(n_jobs, (size_t)std::max(ctx->register_workers, 1)) <= 1;
    static const bool own_warp_stream = ab_dev_env("AB_NO_WARP_STREAM") == nullptr;
    // The warps run at LOW stream priority (round 6): they need nothing but a fitted transform and 4.35 ms of f64 issue slots, while
    // every estimate kernel is on some group's chain to its fit.  Same box, interleaved, three runs each: 10.61 / 10.72 / 10.79 ms
    // per step with the warp stream at normal priority, 10.40 / 10.49 / 10.37 at low (profiles/r06_priority_ab.txt); a HIGH-priority
    // tile stream made it worse (10.82 / 10.78 / 10.92), both together no better than low warps alone.
    if (!inline_run && aligned && own_warp_stream) {
        if (!ctx->warp_stream) {
        const hipError_t stream_rc = ab_stream_create_masked(ctx, &ctx->warp_stream, AB_DEV_NAME("AB_WARP_CU_MASK"), AB_DEV_NAME("AB_WARP_PRIO"), 1);  // (outside AB_HIP: its message would carry the developer variables' names)
        AB_HIP(ctx, stream_rc);
    }
        warp_stream = ctx->warp_stream;
    }
    if (inline_run) {  // the reference first, on ctx
        int rc = prepare_reference();
        rt.publish(rc, rc == AB_OK && rt.stars.size() >= kMinMatchesRigid);
        if (rc == AB_OK) rc = ab_parallel_frames(ctx, n_jobs, "registration", job);
        if (pipe.on) (void)hipStreamSynchronize(ctx->aux_stream);  // (an error or a cancel may leave tile launches in flight: they read the caller's frames)
        return rc;
    }
    AB_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's frames are complete before any other stream reads them
    // the reference frame's detection and triangle table run on the caller's context, on one more thread of the pool, while the
    // workers already detect their targets (they block on the table only before matching)
    int prep_rc = AB_OK;
    const std::function<void()> prep = [&]() {
        prep_rc = prepare_reference();
        rt.publish(prep_rc, prep_rc == AB_OK && rt.stars.size() >= kMinMatchesRigid);
    };
    const int rc = ab_parallel_frames(ctx, n_jobs, "registration", job, /*drain_caller_stream=*/false, &prep);
    ab_upload_trace("workers joined, jobs", (long)n_jobs);
    if (warp_stream) (void)hipStreamSynchronize(warp_stream);
    ab_upload_trace("warp stream drained", 0);
    if (!landed) ab_upload_trace("registration returns", (long)n);  // the aligned frames are complete when this call returns
    if (pipe.on) (void)hipStreamSynchronize(ctx->aux_stream);  // (an error or a cancel may leave tile launches in flight: they read the caller's frames)
    return prep_rc != AB_OK ? prep_rc : rc;
}

int ab_align_channel_affine_device(ab_ctx *ctx, const float *ref, const float *tgt, int64_t rows, int64_t cols, int num_threads,
                                   ab_affine_align_result *out) {
    return ab_register_frames_device(ctx, ref, &tgt, 1, rows, cols, num_threads, out, nullptr, nullptr, 0, -1);
}

extern "C" {

int ab_align_channel_affine(ab_ctx *ctx, const ab_plane *reference, const ab_plane *target, int num_threads,
                            ab_affine_align_result *out) try {
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
} AB_CATCH(ctx)

int ab_register_frames(ab_ctx *ctx, const ab_plane *reference, const ab_plane *targets, size_t n, int num_threads,
                       ab_affine_align_result *out) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, reference && out && (targets || n == 0), "null argument");
    for (size_t i = 0; i < n; ++i)
        AB_CHECK(ctx, targets[i].rows == reference->rows && targets[i].cols == reference->cols,
                 "align_channel_affine takes two planes of the same dims (%lldx%lld vs %lldx%lld)", (long long)reference->rows,
                 (long long)reference->cols, (long long)targets[i].rows, (long long)targets[i].cols);
    StagedPlane r;
    AB_TRY(ab_stage_in(ctx, reference, &r));
    std::vector<StagedPlane> st(n);
    std::vector<const float *> ptrs(n);
    int rc = AB_OK;
    size_t staged = 0;
    for (; staged < n && rc == AB_OK; ++staged) {
        rc = ab_stage_in(ctx, &targets[staged], &st[staged]);
        if (rc != AB_OK) break;
        ptrs[staged] = st[staged].dptr;
    }
    if (rc == AB_OK) rc = ab_register_frames_device(ctx, r.dptr, ptrs.data(), n, r.rows, r.cols, num_threads, out, nullptr, nullptr, 0, -1);
    for (size_t i = 0; i < staged && i < n; ++i) ab_stage_release(ctx, &st[i]);
    ab_stage_release(ctx, &r);
    return rc;
} AB_CATCH(ctx)

int ab_align_pairs_affine(ab_ctx *ctx, const ab_plane *reference, const ab_plane *targets, size_t n, int num_threads,
                          ab_affine_align_result *out, ab_plane_mut *aligned) try {
    if (!ctx) return AB_ERR_INVALID;
    AB_CHECK(ctx, reference && out && aligned && (targets || n == 0), "null argument");
    std::vector<const float *> ptrs(n);
    std::vector<float *> outs(n);
    size_t n_host = 0;
    for (size_t i = 0; i < n; ++i) {
        AB_CHECK(ctx, aligned[i].on_device, "align_pairs writes device-resident planes");
        AB_CHECK(ctx, targets[i].rows == reference->rows && targets[i].cols == reference->cols && aligned[i].rows == reference->rows &&
                          aligned[i].cols == reference->cols,
                 "align_pairs: target %zu / its output differ from the reference's dims", i);
        AB_CHECK(ctx, targets[i].data != aligned[i].data, "warp_image cannot run in place");
        ptrs[i] = targets[i].data;
        outs[i] = aligned[i].data;
        if (!targets[i].on_device) ++n_host;
    }
    if (n_host == 0 && reference->on_device)
        return ab_register_frames_device(ctx, reference->data, ptrs.data(), n, reference->rows, reference->cols, num_threads, out, outs.data(), nullptr, 0, -1);
    // Frames held by the HOST (as the application holds them: Array2<f32> in the image cache, calibration.rs:306-315): every host
    // frame is copied into an HBM staging area on the context's upload stream, an event behind each, all enqueued before anything
    // else; the registration's pipeline waits for a frame's event, not for the batch, so the link is busy from the first byte to the
    // last and what remains after the last byte is one group's registration.  (Pinned host memory keeps the copies asynchronous;
    // pageable memory is staged by the runtime copy by copy -- same result, the call then waits in the enqueue loop.)
    AB_HIP(ctx, hipSetDevice(ctx->device));
    const size_t plane_bytes = (size_t)reference->rows * (size_t)reference->cols * sizeof(float);
    const size_t slots = n_host + (reference->on_device ? 0 : 1);
    const size_t stride = (plane_bytes + 255) & ~(size_t)255;
    if (slots * stride > ctx->upload_bytes) {
        if (ctx->upload_buf) {
            AB_HIP(ctx, hipDeviceSynchronize());
            AB_HIP(ctx, hipFree(ctx->upload_buf));
            ctx->upload_buf = nullptr;
            ctx->upload_bytes = 0;
        }
        AB_HIP(ctx, hipMalloc(&ctx->upload_buf, slots * stride));
        ctx->upload_bytes = slots * stride;
    }
    // (a queue of its own pool: 63 copies are enqueued up front, each a barrier packet that waits for its DMA -- on a queue shared
    // with worker streams they held a quarter of the groups back until the LAST frame had landed: measured, 8 ms after the last byte)
    if (!ctx->upload_stream) {
        const hipError_t stream_rc = ab_stream_create_masked(ctx, &ctx->upload_stream, nullptr, AB_DEV_NAME("AB_UPLOAD_PRIO"), 1);  // (outside AB_HIP: its message would carry the developer variables' names)
        AB_HIP(ctx, stream_rc);
    }
    while (ctx->upload_events.size() < slots) {
        hipEvent_t e;
        AB_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->upload_events.push_back(e);
    }
    // whatever leaves this function, the copies (they read the caller's memory) have drained first
    struct UploadDrain {
        ab_ctx *c;
        ~UploadDrain() { (void)hipStreamSynchronize(c->upload_stream); }
    } upload_drain{ctx};
    size_t slot = 0;
    const float *ref_dev = reference->data;
    if (!reference->on_device) {  // the reference first: every frame's matching needs it
        float *d = (float *)((char *)ctx->upload_buf + slot * stride);
        if (plane_bytes) AB_HIP(ctx, hipMemcpyAsync(d, reference->data, plane_bytes, hipMemcpyHostToDevice, ctx->upload_stream));
        AB_HIP(ctx, hipStreamSynchronize(ctx->upload_stream));
        ref_dev = d;
        ++slot;
    }
    // AB_UPLOAD_TRACE=1 (developer knob): how long the copies took under the registration's load, and what was left after the last
    static const bool up_trace = ab_dev_env("AB_UPLOAD_TRACE") != nullptr;
    hipEvent_t tr0 = nullptr, tr1 = nullptr;
    const auto t_call = std::chrono::steady_clock::now();
    ab_trace_t0() = t_call;
    if (up_trace) {
        AB_HIP(ctx, hipEventCreate(&tr0));
        AB_HIP(ctx, hipEventCreate(&tr1));
        AB_HIP(ctx, hipEventRecord(tr0, ctx->upload_stream));
    }
    std::vector<hipEvent_t> landed(n, nullptr);
    for (size_t i = 0; i < n; ++i) {
        if (targets[i].on_device) continue;
        float *d = (float *)((char *)ctx->upload_buf + slot * stride);
        if (plane_bytes) AB_HIP(ctx, hipMemcpyAsync(d, targets[i].data, plane_bytes, hipMemcpyHostToDevice, ctx->upload_stream));
        AB_HIP(ctx, hipEventRecord(ctx->upload_events[slot], ctx->upload_stream));
        landed[i] = ctx->upload_events[slot];
        ptrs[i] = d;
        ++slot;
    }
    if (up_trace) AB_HIP(ctx, hipEventRecord(tr1, ctx->upload_stream));
    const double enq_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
    const int rc = ab_register_frames_device(ctx, ref_dev, ptrs.data(), n, reference->rows, reference->cols, num_threads, out, outs.data(),
                                             n_host ? landed.data() : nullptr, 0, -1);
    ab_upload_trace("registration returned", (long)n);
    if (up_trace) {
        const double call_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
        float up_ms = 0.0f;
        (void)hipEventSynchronize(tr1);
        (void)hipEventElapsedTime(&up_ms, tr0, tr1);
        fprintf(stderr, "[ab upload] %zu host frames: enqueue %.2f ms, copies %.2f ms (%.1f GB/s), call %.2f ms\n", n_host, enq_ms, up_ms,
                (double)n_host * (double)plane_bytes / (up_ms * 1e-3) / 1e9, call_ms);
        (void)hipEventDestroy(tr0);
        (void)hipEventDestroy(tr1);
    }
    return rc;
} AB_CATCH(ctx)

// the host geometry alone, on given centroids (x, y pairs): returns AB_OK and *found = 0/1
int ab_affine_from_stars(const double *ref_xy, size_t n_ref, const double *tgt_xy, size_t n_tgt, int64_t rows, int64_t cols,
                         int num_threads, ab_affine_align_result *out, int *found) try {
    if ((!ref_xy && n_ref) || (!tgt_xy && n_tgt) || !out || !found) return AB_ERR_INVALID;
    std::vector<Pt> rs(n_ref), ts(n_tgt);
    for (size_t i = 0; i < n_ref; ++i) rs[i] = {ref_xy[2 * i], ref_xy[2 * i + 1]};
    for (size_t i = 0; i < n_tgt; ++i) ts[i] = {tgt_xy[2 * i], tgt_xy[2 * i + 1]};
    *found = affine_from_stars(rs, ts, rows, cols, num_threads, out) ? 1 : 0;
    return AB_OK;
} AB_CATCH_NOCTX

}  // extern "C"
