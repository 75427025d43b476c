cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_stats_stf.py -m gpu -x -q < /dev/null 2>&1 | tail -n 5 > gpurun_out/r03y_pytest_stats.log
timeout 300 python tools/time_stats.py < /dev/null 2>&1 | tail -n 5 > gpurun_out/r03y_time_stats.txt
