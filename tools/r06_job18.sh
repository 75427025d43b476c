#!/bin/bash
# round 6: 65 .. 128 frames with two waves per SIMD (256 registers, 61-84 spilled); the two-lane kernel with the v_med3 cross step
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_stack.py -m gpu -x -q > gpurun_out/r06r_tests.log 2>&1; echo "stack tests rc=$?" > gpurun_out/r06r_rc.txt
tail -3 gpurun_out/r06r_tests.log
( echo "# tools/time_stack_deep.py, 4096^2: 65 .. 128 frames at two waves per SIMD; 129 .. 256 with the v_med3 cross step"
  N_LIST=65,96,100,128,129,160,200,256 timeout 900 python tools/time_stack_deep.py
  N_LIST=320,512 timeout 900 python tools/time_stack_deep.py
) > gpurun_out/r06r_deep.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06r_deep.txt
timeout 600 python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r06r_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r06r_bench.json')); print(d['ms_per_step'], d['roofline']['frac'], d['config']['stage_ms'])"
