#!/bin/bash
# developer job: the round-5 GPU checks, every step bounded and logged in full (a hung step must not eat the others' evidence)
mkdir -p gpurun_out
PT="python -m pytest -m gpu -x -v --timeout=240 --timeout-method=thread -p no:cacheprovider"
timeout 600 $PT tests/test_gpu_stack.py -k "more_than or workgroup_per_pixel" > gpurun_out/r05b_stack.txt 2>&1; echo "stack rc=$?" >> gpurun_out/r05b_rc.txt
timeout 600 $PT tests/test_gpu_batch.py -k "more_than_512 or workgroup_per_pixel or pipeline_errors" > gpurun_out/r05b_batch.txt 2>&1; echo "batch rc=$?" >> gpurun_out/r05b_rc.txt
timeout 300 $PT tests/test_gpu_extras.py -k create_master > gpurun_out/r05b_extras.txt 2>&1; echo "extras rc=$?" >> gpurun_out/r05b_rc.txt
timeout 900 $PT tests/test_gpu_detect_affine.py > gpurun_out/r05b_detect.txt 2>&1; echo "detect rc=$?" >> gpurun_out/r05b_rc.txt
timeout 600 $PT tests/test_gpu_multirank.py -k "bands or stale or source_rows" > gpurun_out/r05b_multirank.txt 2>&1; echo "multirank rc=$?" >> gpurun_out/r05b_rc.txt
SIDE=2048 N_LIST=513,1024 timeout 300 python tools/time_stack_deep.py > gpurun_out/r05b_deep.txt 2>&1; echo "deep rc=$?" >> gpurun_out/r05b_rc.txt
timeout 600 $PT tests/test_gpu_concurrent.py > gpurun_out/r05b_concurrent.txt 2>&1; echo "concurrent rc=$?" >> gpurun_out/r05b_rc.txt
cat gpurun_out/r05b_rc.txt
for f in stack batch extras detect multirank concurrent; do echo "== $f"; tail -5 gpurun_out/r05b_$f.txt; done
cat gpurun_out/r05b_deep.txt
