cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 build/tile_bench 0 2>&1 | cut -c1-330 | head -5 > gpurun_out/r03v_tile_bench.txt
timeout 900 python -m pytest tests/test_gpu_tile_stats.py tests/test_gpu_detect_affine.py -m gpu -x -q 2>&1 | tail -n 5 > gpurun_out/r03v_pytest.log
timeout 300 python tools/time_register.py > gpurun_out/r03v_time_register.txt 2>&1
