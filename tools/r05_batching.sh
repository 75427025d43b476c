#!/bin/bash
# developer job: the records form of the tile labelling (label_tile_body<true, true> + comp_merge) against the pixel-list chain
mkdir -p gpurun_out
timeout 120 build/label_bench 4 > gpurun_out/r05y_label_bench.txt 2>&1; grep -v "variant [1-6]" gpurun_out/r05y_label_bench.txt
timeout 900 python -m pytest tests/test_gpu_detect_affine.py -m gpu -x -q --timeout=600 --timeout-method=thread -p no:cacheprovider > gpurun_out/r05y_tests.txt 2>&1; tail -15 gpurun_out/r05y_tests.txt
out=gpurun_out/r05y_recs.txt; : > $out
for rep in 1 2 3; do
  for cfg in "AB_NOOP=1" "AB_DETECT_NO_RECS=1"; do
    env $cfg REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
    env $cfg NOALIGN=1 REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
  done
done
cut -c1-150 $out
