#!/bin/bash
# developer job: the round's end figures -- full GPU suite, bench + rocprofv3 summaries (kernel trace, PMC fetch / write), the other
# configurations, the multi-rank code paths
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06z_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r06z_rc.txt; tail -3 gpurun_out/r06z_tests.log
bash tools/profile_bench.sh r06final 7 > gpurun_out/r06final_profile.log 2>&1; tail -3 gpurun_out/r06final_profile.log
timeout 600 python bench.py > gpurun_out/r06z_bench.json 2> gpurun_out/r06z_bench.err; echo "bench rc=$?" >> gpurun_out/r06z_rc.txt
for c in C1 C3 C5; do
  timeout 900 python bench.py --config $c > gpurun_out/r06z_bench_$c.json 2> gpurun_out/r06z_bench_$c.err; echo "$c rc=$?" >> gpurun_out/r06z_rc.txt
done
timeout 600 python bench.py --force-sharded --no-cpu-baseline > gpurun_out/r06z_bench_sharded.json 2> gpurun_out/r06z_sharded.err; echo "sharded rc=$?" >> gpurun_out/r06z_rc.txt
timeout 600 python bench.py --force-sharded --mode rowband --no-cpu-baseline > gpurun_out/r06z_bench_rowband.json 2> gpurun_out/r06z_rowband.err; echo "rowband rc=$?" >> gpurun_out/r06z_rc.txt
for m in frames rowband; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --host-staged --mode $m --steps 3 --warmup 1 --no-cpu-baseline \
     > gpurun_out/r06z_bench_hoststaged_8ranks_$m.json 2> gpurun_out/r06z_hoststaged_$m.err; echo "hoststaged $m rc=$?" >> gpurun_out/r06z_rc.txt
done
timeout 600 python bench.py --host-planes --no-cpu-baseline > gpurun_out/r06z_bench_host_planes.json 2> gpurun_out/r06z_host_planes.err; echo "host planes rc=$?" >> gpurun_out/r06z_rc.txt
cat gpurun_out/r06z_rc.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06z_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(f, d["ms_per_step"], d.get("roofline", {}).get("frac"), d["config"].get("stage_ms"), d.get("roofline_step", {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
ls gpurun_out/prof_r06final
