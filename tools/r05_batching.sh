#!/bin/bash
mkdir -p gpurun_out
timeout 120 build/label_bench 4 > gpurun_out/r05aa_label_bench.txt 2>&1; grep -v "variant [1-6]" gpurun_out/r05aa_label_bench.txt
timeout 900 python -m pytest tests/test_gpu_detect_affine.py -m gpu -x -q --timeout=600 --timeout-method=thread -p no:cacheprovider > gpurun_out/r05aa_tests.txt 2>&1; tail -5 gpurun_out/r05aa_tests.txt
out=gpurun_out/r05aa_select.txt; : > $out
for rep in 1 2; do
    REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
    NOALIGN=1 REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
done
cut -c1-150 $out
