#!/bin/bash
# round 6: four lanes per pixel for 257 .. 512 frames (stack_quad.hip): parity, then times at 4096^2
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
timeout 1500 python -m pytest tests/test_gpu_stack.py -m gpu -x -q > gpurun_out/r06q_tests.log 2>&1; echo "stack tests rc=$?" > gpurun_out/r06q_rc.txt
tail -5 gpurun_out/r06q_tests.log
( echo "# tools/time_stack_deep.py, 4096^2, the default engine (round 6: two lanes per pixel for 129 .. 256 frames, four for 257 .. 512)"
  N_LIST=257,264,320,384,392,448,512 timeout 900 python tools/time_stack_deep.py
  echo "## pixels handed to the list pass (AB_TRACE=1)"
  AB_TRACE=1 N_LIST=320,512 timeout 900 python tools/time_stack_deep.py 2>&1 | grep -E "two-lane|frames x" | awk '!seen[$0]++'
  echo "## two lanes per pixel with 256 samples each (developer build, AB_STACK_NO_QUAD=1)"
  AB_LIB_PATH=$DEV AB_STACK_NO_QUAD=1 N_LIST=320,512 timeout 600 python tools/time_stack_deep.py
) > gpurun_out/r06q_deep.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06q_deep.txt
