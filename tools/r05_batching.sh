#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_stack.py tests/test_gpu_extras.py -m gpu -x -q --timeout=900 --timeout-method=thread -p no:cacheprovider > gpurun_out/r05ar_tests.txt 2>&1; tail -5 gpurun_out/r05ar_tests.txt | cut -c1-300
out=gpurun_out/r05ar_quad.txt; : > $out
for cfg in "AB_NOOP=1" "AB_STACK_PAIR=1"; do
  echo "## $cfg" >> $out
  env $cfg N_LIST=257,320,400,512 timeout 600 python tools/time_stack_deep.py 2>&1 | grep -v "^/opt" >> $out
done
cat $out
