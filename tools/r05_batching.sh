#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r05ai_slices.txt; : > $out
for rep in 1 2; do
  for cfg in "AB_VOTE_SLICES=8" "AB_VOTE_SLICES=4" "AB_VOTE_SLICES=2" "AB_VOTE_SLICES=1"; do
    env $cfg REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
    env $cfg NOALIGN=1 REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
  done
done
cut -c1-130 $out
timeout 600 python -m pytest tests/test_gpu_detect_affine.py -m gpu -x -q --timeout=600 --timeout-method=thread -p no:cacheprovider 2>&1 | tail -2
