#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_resample.py tests/test_gpu_detect_affine.py -m gpu -x -q --timeout=600 --timeout-method=thread -p no:cacheprovider > gpurun_out/r05ag_tests.txt 2>&1; tail -5 gpurun_out/r05ag_tests.txt
out=gpurun_out/r05ag_warp.txt; : > $out
for rep in 1 2 3; do
  for cfg in "AB_NOOP=1" "AB_WARP_DIRECT=1"; do
    env $cfg REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
  done
done
cut -c1-150 $out
timeout 200 python tools/time_warp.py 2>&1 | tail -5
AB_WARP_DIRECT=1 timeout 200 python tools/time_warp.py 2>&1 | tail -5
