#!/usr/bin/env python3
"""Developer tool: what the GPU does during one ab_align_pairs_affine call, from a rocprofv3 --kernel-trace rocpd database of
tools/time_register.py.  The kernel timeline is cut into bursts at idle gaps > 150 us (the calls are separated by host
synchronisation); the bursts that hold a tile kernel are the registration calls.  For each: span, union busy time, concurrency
histogram (share of the span with k kernels in flight), and kernel time by name.
Usage: python tools/register_timeline.py trace_results.db"""
import collections
import sqlite3
import sys


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][:40]


def main(path):
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    s_col = "start" if "start" in cols else "start_time"
    e_col = "end" if "end" in cols else "end_time"
    rows = sorted(con.execute(f"select name, {s_col}, {e_col} from kernels").fetchall(), key=lambda r: r[1])
    bursts, cur, cur_end = [], [], None
    for n, s, e in rows:
        if cur_end is not None and s - cur_end > 150_000:
            bursts.append(cur)
            cur = []
        cur.append((n, s, e))
        cur_end = e if cur_end is None or not cur[:-1] else max(cur_end, e)
    bursts.append(cur)
    big = [b for b in bursts if any("tile_background_stream" in n for n, _, _ in b)]
    print(f"{len(rows)} kernels, {len(bursts)} bursts, {len(big)} registration calls")
    for b in big:  # one line per call: where a slow call loses its time (an idle GPU = the host)
        t0, t1 = min(s for _, s, _ in b), max(e for _, _, e in b)
        ev = sorted([(s, 1) for _, s, _ in b] + [(e, -1) for _, _, e in b])
        depth, last, idle, gap = 0, t0, 0, 0
        for t, d in ev:
            if depth == 0:
                idle += t - last
                gap = max(gap, t - last)
            last, depth = t, depth + d
        print(f"call: span {(t1 - t0) / 1e6:7.3f} ms  idle {idle / 1e6:6.3f} ms  longest gap {gap / 1e6:6.3f} ms  kernels {len(b)}")
    for b in big[-3:]:
        t0, t1 = min(s for _, s, _ in b), max(e for _, _, e in b)
        ev = sorted([(s, 1) for _, s, _ in b] + [(e, -1) for _, _, e in b])
        hist, depth, last = collections.Counter(), 0, t0
        for t, d in ev:
            hist[depth] += t - last
            last, depth = t, depth + d
        # which kernel is it when exactly ONE is in flight
        ev2 = sorted([(s, 1, n) for n, s, _ in b] + [(e, -1, n) for n, _, e in b])
        alone, live, last2 = collections.Counter(), collections.Counter(), t0
        for t, d, n in ev2:
            if sum(live.values()) == 1:
                alone[short(next(k for k, v in live.items() if v > 0))] += t - last2
            last2 = t
            live[n] += d
        span = t1 - t0
        print("  alone in flight (ms): " + ", ".join(f"{k} {v / 1e6:.2f}" for k, v in alone.most_common(8)))
        tot = sum(e - s for _, s, e in b)
        print(f"\nburst: {len(b)} kernels, span {span / 1e6:.3f} ms, kernel time {tot / 1e6:.3f} ms, mean concurrency {tot / span:.2f}")
        print("  share of the span with k kernels in flight: " + "  ".join(f"{k}:{v / span:.3f}" for k, v in sorted(hist.items())))
        by = collections.defaultdict(lambda: [0, 0])
        for n, s, e in b:
            key = short(n)
            by[key][0] += e - s
            by[key][1] += 1
        for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:18]:
            print(f"  {k:48s} {c:5d} launches {t / 1e6:8.3f} ms  avg {t / c / 1e3:7.1f} us")
        # when the warps start (a warp needs its frame's fitted transform: the call ends ~ first warp + the warps' durations)
        ws = sorted((s - t0, e - s) for n, s, e in b if "warp_kernel" in n)
        if ws:
            pick = [0, 7, 15, 31, 47, len(ws) - 1]
            print("  warp k starts at (ms): " + "  ".join(f"{k + 1}:{ws[k][0] / 1e6:.2f}" for k in pick if k < len(ws)) +
                  f"   durations (us) first 16: {sum(d for _, d in ws[:16]) / 16e3:.0f}  last 16: {sum(d for _, d in ws[-16:]) / 16e3:.0f}")
        fam = collections.defaultdict(lambda: [1 << 62, 0])
        for n, s, e in b:
            k = short(n)
            fam[k][0] = min(fam[k][0], s - t0)
            fam[k][1] = max(fam[k][1], e - t0)
        print("  first start .. last end (ms): " + ", ".join(f"{k} {v[0] / 1e6:.2f}..{v[1] / 1e6:.2f}" for k, v in sorted(fam.items(), key=lambda kv: kv[1][0])
                                                              if k in ("subsample_many_kernel", "percentiles_many_reg_kernel", "tile_background_stream_kernel", "label_bgtile_many_kernel",
                                                                       "comp_select_many_kernel", "tri_vote_wide_many_kernel", "votes_reduce_many_kernel", "warp_kernel")))
        # the tail: what runs in the last 1.5 ms
        tail = [(n, s, e) for n, s, e in b if e > t1 - 1_500_000]
        print("  last 1.5 ms: " + ", ".join(f"{k}x{c}" for k, c in collections.Counter(short(n) for n, _, _ in tail).most_common(8)))


if __name__ == "__main__":
    main(sys.argv[1])
