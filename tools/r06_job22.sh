#!/bin/bash
# round 6: the four-lane kernel with XCD-aware pixel groups; the multi-lane passes under odd clip settings
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_stack.py -m gpu -x -q -k "two_lane or two_lanes or deep" > gpurun_out/r06v_tests.log 2>&1; echo "stack tests rc=$?" > gpurun_out/r06v_rc.txt
tail -3 gpurun_out/r06v_tests.log
( echo "# tools/time_stack_deep.py, 4096^2: four lanes per pixel, pixel groups id and id + 8 (same XCD) adjacent"
  N_LIST=257,320,384,448,512 timeout 900 python tools/time_stack_deep.py
  MODE=median N_LIST=320,512 timeout 900 python tools/time_stack_deep.py
) > gpurun_out/r06v_deep.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06v_deep.txt
