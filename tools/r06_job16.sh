#!/bin/bash
# round 6: the two-lane fast pass for 129 .. 512 frames (stack_duo.hip): parity, then times at 4096^2
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
timeout 1500 python -m pytest tests/test_gpu_stack.py -m gpu -x -q > gpurun_out/r06p_tests.log 2>&1; echo "stack tests rc=$?" > gpurun_out/r06p_rc.txt
tail -5 gpurun_out/r06p_tests.log
( echo "# tools/time_stack_deep.py, 4096^2, the default engine (round 6: two lanes per pixel from 129 frames on)"
  N_LIST=128,129,136,160,192,200,224,256 timeout 600 python tools/time_stack_deep.py
  N_LIST=257,320,384,448,512 timeout 900 python tools/time_stack_deep.py
  echo "## pixels handed to the list pass (AB_TRACE=1)"
  AB_TRACE=1 N_LIST=129,200,256,320,512 timeout 900 python tools/time_stack_deep.py 2>&1 | grep -E "two-lane|frames x"
  echo "## round 5's one-lane 256-sample kernel (developer build, AB_STACK_NO_DUO=1)"
  AB_LIB_PATH=$DEV AB_STACK_NO_DUO=1 N_LIST=160,200,256 timeout 600 python tools/time_stack_deep.py
) > gpurun_out/r06p_deep.txt 2>&1
cat gpurun_out/r06p_deep.txt
