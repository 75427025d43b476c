#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_detect_affine.py -m gpu -x -q --timeout=600 --timeout-method=thread -p no:cacheprovider 2>&1 | tail -2
out=gpurun_out/r05an_pct.txt; : > $out
for rep in 1 2 3; do
  for cfg in "AB_NOOP=1" "AB_PCT_TWO_KERNELS=1"; do
    env $cfg REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
    env $cfg NOALIGN=1 REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
  done
done
cut -c1-140 $out
AB_UPLOAD_TRACE=1 REPS=1 timeout 100 python tools/time_register.py 2>&1 | grep "percentiles joined" | tail -2
