#!/bin/bash
# round 6: eight lanes per pixel for 513 .. 1024 frames (stack_quad.hip, L = 8)
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
timeout 2400 python -m pytest tests/test_gpu_stack.py -m gpu -x -q > gpurun_out/r06y_tests.log 2>&1; echo "stack tests rc=$?" > gpurun_out/r06y_rc.txt
tail -5 gpurun_out/r06y_tests.log
( echo "# tools/time_stack_deep.py: 513 .. 1024 frames, eight lanes per pixel (default) against the wave-per-pixel kernel (developer build, AB_STACK_NO_OCTO=1)"
  SIDE=2048 N_LIST=513,640,768,1024 timeout 900 python tools/time_stack_deep.py
  SIDE=2048 MODE=median N_LIST=513,1024 timeout 900 python tools/time_stack_deep.py
  SIDE=4096 N_LIST=513,1024 timeout 900 python tools/time_stack_deep.py
  echo "## pixels handed to the list pass (AB_TRACE=1)"
  SIDE=2048 AB_TRACE=1 N_LIST=1024 timeout 900 python tools/time_stack_deep.py 2>&1 | grep -E "two-lane|frames x" | awk '!seen[$0]++'
  echo "## AB_STACK_NO_OCTO=1 (developer build)"
  AB_LIB_PATH=$DEV AB_STACK_NO_OCTO=1 SIDE=2048 N_LIST=513,1024 timeout 900 python tools/time_stack_deep.py
) > gpurun_out/r06y_deep.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06y_deep.txt
