#!/bin/bash
# round 6: the median combine and the exact engine of 129 .. 256 frames through stack_pair.hip's kernel with 128 samples per lane
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
timeout 1500 python -m pytest tests/test_gpu_stack.py -m gpu -x -q > gpurun_out/r06t_tests.log 2>&1; echo "stack tests rc=$?" > gpurun_out/r06t_rc.txt
tail -3 gpurun_out/r06t_tests.log
( echo "# tools/time_stack_deep.py, 4096^2: median combine and AB_STACK_EXACT=1 of 129 .. 256 frames, two lanes per pixel (stack_pair.hip, 128 samples per lane)"
  MODE=median N_LIST=100,128,160,200,256,320,512 timeout 900 python tools/time_stack_deep.py
  AB_STACK_EXACT=1 N_LIST=160,256 timeout 900 python tools/time_stack_deep.py
  echo "## round 5's routes (developer build, AB_STACK_NO_DUO=1): one lane with 256 samples for the median, a wave per pixel for the exact engine"
  AB_LIB_PATH=$DEV AB_STACK_NO_DUO=1 MODE=median N_LIST=160,200,256 timeout 900 python tools/time_stack_deep.py
  AB_LIB_PATH=$DEV AB_STACK_NO_DUO=1 AB_STACK_EXACT=1 N_LIST=160,256 timeout 900 python tools/time_stack_deep.py
) > gpurun_out/r06t_deep.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06t_deep.txt
