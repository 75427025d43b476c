cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_background.py tests/test_gpu_masked.py tests/test_gpu_compose.py tests/test_gpu_extras.py tests/test_gpu_full_size.py -m gpu -x -q < /dev/null 2>&1 | head -n 4 > gpurun_out/r03ag_pytest.log
timeout 300 python tools/time_c5.py < /dev/null > gpurun_out/r03ag_c5.txt 2>&1
