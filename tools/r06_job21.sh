#!/bin/bash
# round 6: the median combine of 129 .. 512 frames through the fast passes; then the whole GPU suite on this tree
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_stack.py -m gpu -x -q > gpurun_out/r06u_tests.log 2>&1; echo "stack tests rc=$?" > gpurun_out/r06u_rc.txt
tail -3 gpurun_out/r06u_tests.log
( echo "# tools/time_stack_deep.py, 4096^2: the median combine through the fast passes (stack_duo.hip / stack_quad.hip, MEDIAN instances)"
  MODE=median N_LIST=129,160,200,256,320,512 timeout 900 python tools/time_stack_deep.py
) > gpurun_out/r06u_deep.txt 2>&1
grep -v amdgpu.ids gpurun_out/r06u_deep.txt
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r06u_all_tests.log 2>&1; echo "all gpu tests rc=$?" >> gpurun_out/r06u_rc.txt
tail -3 gpurun_out/r06u_all_tests.log
cat gpurun_out/r06u_rc.txt
