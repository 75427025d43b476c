cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_phasecorr.py tests/test_golden.py tests/test_gpu_full_size.py tests/test_gpu_stack.py -m gpu -x -q < /dev/null 2>&1 | head -n 4 > gpurun_out/r03am_pytest.log
timeout 600 python bench.py --config C1 < /dev/null 2>/dev/null | tail -n 1 > gpurun_out/r03am_C1.json
