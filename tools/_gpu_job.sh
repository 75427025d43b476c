cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_stats_stf.py -m gpu -x -q 2>&1 | tail -n 5 > gpurun_out/r03w_pytest_stats.log
AB_STATS_TIMING=1 timeout 300 python tools/time_stats.py 2>&1 | tail -n 7 > gpurun_out/r03w_time_stats.txt
