#!/bin/bash
# round 6: the > 512-frame LDS-staged kernel with XCD-aware group order
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_stack.py -m gpu -x -q -k "more_than_512" > gpurun_out/r06x_tests.log 2>&1; echo "stack tests rc=$?" > gpurun_out/r06x_rc.txt
tail -2 gpurun_out/r06x_tests.log
( echo "# tools/time_stack_deep.py, 2048^2: 513 .. 4096 frames, groups of one XCD's workgroups adjacent (was 43.7 / 48.7 ms for 513 / 1024)"
  SIDE=2048 N_LIST=513,1024,2048 timeout 900 python tools/time_stack_deep.py
  SIDE=1024 N_LIST=2100,4096 timeout 900 python tools/time_stack_deep.py
) > gpurun_out/r06x_deep.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06x_deep.txt
