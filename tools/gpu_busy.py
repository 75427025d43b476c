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
stack = [e for e in ev if "stack_sigma_clip_kernel<64" in e[2] and ", 1, 0>" in (e[2] + ", 1, 0>")]
stack = [e for e in ev if "stack_sigma_clip_kernel" in e[2] and e[1] - e[0] > 500_000]
t_end = stack[-2][0]
t_beg = t_end - int(win * 1e6)
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
print(f"window {win} ms before the last-but-one stack: {len(seg)} kernels, busy union {busy / 1e6:.2f} ms, sum of durations {sum(e - s for s, e, _, _ in seg) / 1e6:.2f} ms, queues {len(set(q for *_, q in seg))}")
d = collections.defaultdict(list)
for s, e, n, _ in seg:
    d[n].append(e - s)
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print(f"{n:62s} n={len(v):4d} sum={sum(v) / 1e6:7.2f} ms avg={sum(v) / len(v) / 1e3:7.1f} us min={min(v) / 1e3:6.1f}")
# concurrency histogram: time spent with k kernels in flight
pts = sorted([(s, 1) for s, e, _, _ in seg] + [(e, -1) for s, e, _, _ in seg])
k, last, hist = 0, t_beg, collections.Counter()
for t, dlt in pts:
    hist[k] += t - last
    last = t
    k += dlt
print("time with k kernels in flight (ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
