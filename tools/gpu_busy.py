"""Developer tool: from a rocprofv3 kernel trace (csv) of bench.py, how busy was the GPU during the registration stage?
   python tools/gpu_busy.py <kernel_trace.csv> [window_ms]
Prints, for the window before the last-but-one stack launch: the union of the kernel intervals (GPU busy time), the sum of the
kernel durations (> union when kernels overlap) and the per-kernel sums -- 'is the stage GPU-bound or host-bound'."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 28.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60], r.get("Queue_Id", "")) for r in rows)
starts = [e for e in ev if "subsample_many_kernel" in e[2]]      # a registration batch begins with its frames' subsamples
if starts:
    t_beg = starts[-1][0]
else:                                                             # (older builds: one tile launch per frame)
    tiles = [e for e in ev if "tile_background" in e[2]][-63:]
    t_beg = min(e[0] for e in tiles) - 300_000
t_end = min(e[0] for e in ev if "stack_sigma_clip_kernel" in e[2] and e[0] > t_beg)
win = (t_end - t_beg) / 1e6
seg = [e for e in ev if e[0] >= t_beg and e[1] <= t_end]
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in seg:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"registration stage of the last step, {win:.2f} ms: {len(seg)} kernels, busy union {busy / 1e6:.2f} ms, sum of durations {sum(e - s for s, e, _, _ in seg) / 1e6:.2f} ms, queues {len(set(q for *_, q in seg))}")
d = collections.defaultdict(list)
for s, e, n, _ in seg:
    d[n].append(e - s)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print(f"{n:62s} n={len(v):4d} sum={sum(v) / 1e6:7.2f} ms avg={sum(v) / len(v) / 1e3:7.1f} us min={min(v) / 1e3:6.1f}")
# concurrency histogram: time spent with k kernels in flight
pts = sorted([(s, 1, n) for s, e, n, _ in seg] + [(e, -1, n) for s, e, n, _ in seg])
last, hist, alone, live = t_beg, collections.Counter(), collections.Counter(), collections.Counter()
for t, dlt, n in pts:
    k = sum(live.values())
    hist[k] += t - last
    if k == 1:
        alone[next(iter(+live))] += t - last
    last = t
    live[n] += dlt
print("time with k kernels in flight (ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
print("running alone (ms):", {n[:28]: round(v / 1e6, 2) for n, v in alone.most_common(8)})
