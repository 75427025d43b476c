cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_detect_affine.py tests/test_gpu_multirank.py -m gpu -x -q < /dev/null 2>&1 | grep -E "passed|failed|error" | tail -n 2 > gpurun_out/r03ao.txt
for i in 1 2 3 4; do timeout 120 python tools/time_register.py < /dev/null 2>&1 | tail -n 1 | cut -c1-70 >> gpurun_out/r03ao.txt; done
