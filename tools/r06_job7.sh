#!/bin/bash
# developer job (round 6): the row-band registration as one call + the 64-workgroup tile fallback: parity, then the multi-rank code paths
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06g_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r06g_rc.txt; tail -4 gpurun_out/r06g_tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r06g_bench.json 2> gpurun_out/r06g_bench.err; echo "bench rc=$?" >> gpurun_out/r06g_rc.txt
timeout 600 python bench.py --force-sharded --no-cpu-baseline > gpurun_out/r06g_bench_sharded.json 2> gpurun_out/r06g_sharded.err; echo "sharded rc=$?" >> gpurun_out/r06g_rc.txt
timeout 600 python bench.py --force-sharded --mode rowband --no-cpu-baseline > gpurun_out/r06g_bench_rowband.json 2> gpurun_out/r06g_rowband.err; echo "rowband rc=$?" >> gpurun_out/r06g_rc.txt
for m in frames rowband; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --host-staged --mode $m --steps 3 --warmup 1 --no-cpu-baseline \
     > gpurun_out/r06g_bench_hoststaged_8ranks_$m.json 2> gpurun_out/r06g_hoststaged_$m.err; echo "hoststaged $m rc=$?" >> gpurun_out/r06g_rc.txt
done
cat gpurun_out/r06g_rc.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06g_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(f, d["ms_per_step"], d.get("roofline", {}).get("frac"), d["config"].get("stage_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -3 gpurun_out/r06g_hoststaged_rowband.err
