#!/bin/bash
# developer job (round 6): full suite (fill-ordering fix, > 512-frame stacks through LDS with tree sums), deep-stack timings, C-configs
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06h_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r06h_rc.txt; tail -4 gpurun_out/r06h_tests.log
( echo "# Round 6: stacks of more than 512 frames at 2048^2 (tools/time_stack_deep.py): groups of 16 / 8 / 4 adjacent pixels staged through LDS,"
  echo "# one sort + the window formula for the MAD, the survivors summed as a tree (AB_STACK_EXACT=1: the oracle's ascending chain)"
  SIDE=2048 N_LIST=513,1024 timeout 600 python tools/time_stack_deep.py 2>&1 | grep -v amdgpu
  echo "## AB_STACK_EXACT=1"
  AB_STACK_EXACT=1 SIDE=2048 N_LIST=513,1024 timeout 600 python tools/time_stack_deep.py 2>&1 | grep -v amdgpu
  echo "## 2100 frames x 1024^2"
  SIDE=1024 N_LIST=2100 timeout 600 python tools/time_stack_deep.py 2>&1 | grep -v amdgpu ) > gpurun_out/r06h_deep.txt
cat gpurun_out/r06h_deep.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r06h_bench.json 2> gpurun_out/r06h_bench.err; echo "bench rc=$?" >> gpurun_out/r06h_rc.txt
for c in C1 C3 C5; do timeout 900 python bench.py --config $c > gpurun_out/r06h_bench_$c.json 2> gpurun_out/r06h_bench_$c.err; echo "$c rc=$?" >> gpurun_out/r06h_rc.txt; done
cat gpurun_out/r06h_rc.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06h_bench*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(f, d["ms_per_step"], d.get("roofline", {}).get("frac"), d["config"].get("stage_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
