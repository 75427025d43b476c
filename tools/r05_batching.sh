#!/bin/bash
# developer job: run-based tile labelling against the pixel-by-pixel form, same box, interleaved
mkdir -p gpurun_out
out=gpurun_out/r05x_label_runs.txt; : > $out
timeout 900 python -m pytest tests/test_gpu_detect_affine.py -m gpu -x -q --timeout=600 --timeout-method=thread -p no:cacheprovider > gpurun_out/r05x_tests.txt 2>&1; tail -3 gpurun_out/r05x_tests.txt
for rep in 1 2 3; do
  for cfg in "AB_NOOP=1" "AB_LABEL_PIXELWISE=1"; do
    env $cfg REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
    env $cfg NOALIGN=1 REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
  done
done
cut -c1-150 $out
