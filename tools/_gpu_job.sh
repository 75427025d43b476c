set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/pmc_stack_sq.sh > gpurun_out/r03g_sq.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/kt.log 2>&1
f=$(ls /tmp/kt/*/kt_kernel_trace.csv /tmp/kt/kt_kernel_trace.csv 2>/dev/null | head -1)
cd $GRAFT_REPO_ROOT
python tools/gpu_busy.py $f > gpurun_out/r03g_gpu_busy.txt 2>&1
python - "$f" <<'PY' > gpurun_out/r03g_timeline.txt 2>&1
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40], r.get("Queue_Id", ""), r.get("Stream_Id","")) for r in rows)
starts = [e for e in ev if "subsample_many_kernel" in e[2]]
t_beg = starts[-1][0]
t_end = min(e[0] for e in ev if "stack_sigma_clip_kernel" in e[2] and e[0] > t_beg)
seg = [e for e in ev if e[0] >= t_beg and e[1] <= t_end]
print("stage", (t_end - t_beg) / 1e6, "ms", len(seg), "kernels")
# per-queue busy
byq = collections.defaultdict(list)
for s, e, n, q, st in seg: byq[q].append((s, e, n))
for q, v in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in v)
    print("queue", q, "kernels", len(v), "busy ms", busy / 1e6)
# first 400 events relative
for s, e, n, q, st in seg[:500]:
    print(f"{(s - t_beg) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{q} s{st} {n}")
PY
