#!/bin/bash
mkdir -p gpurun_out
PT="python -m pytest -m gpu -x -q --timeout=600 --timeout-method=thread -p no:cacheprovider"
timeout 900 $PT tests/test_gpu_detect_affine.py > gpurun_out/r05r_tests.txt 2>&1; echo "tests rc=$?" > gpurun_out/r05r_rc.txt
REPS=3 AB_TRACE=1 timeout 300 python tools/time_register.py 2>&1 | grep -c "redone in full" > gpurun_out/r05r_fallbacks.txt
for rep in 1 2 3; do
  REPS=10 timeout 300 python tools/time_register.py >> gpurun_out/r05r_register.txt 2>&1
done
for rep in 1 2; do timeout 900 python bench.py --config C3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('C3', d['ms_per_step'], d['config']['stage_ms'])" >> gpurun_out/r05r_c3.txt; done
cat gpurun_out/r05r_rc.txt; tail -2 gpurun_out/r05r_tests.txt; echo "fallbacks: $(cat gpurun_out/r05r_fallbacks.txt)"
grep -v "^/opt" gpurun_out/r05r_register.txt | cut -c1-160; cat gpurun_out/r05r_c3.txt
