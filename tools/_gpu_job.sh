cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_phasecorr.py tests/test_golden.py tests/test_gpu_full_size.py tests/test_gpu_stack.py tests/test_gpu_compose.py tests/test_known_answers.py -m gpu -x -q < /dev/null 2>&1 | head -n 6 > gpurun_out/r03ak_pytest.log
timeout 600 python bench.py --config C1 < /dev/null 2>/dev/null | tail -n 1 > gpurun_out/r03ak_C1.json
timeout 300 python tools/time_c5.py < /dev/null 2>&1 | grep -E "process_rgb|spcc" >> gpurun_out/r03ak_pytest.log
