#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_detect_affine.py tests/test_gpu_stack.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/r06k_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r06k_rc.txt; tail -3 gpurun_out/r06k_tests.log
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print("bench", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_sustained"], d["config"]["stage_ms"], d["roofline_step"]["frac"])'; done
cat gpurun_out/r06k_rc.txt
