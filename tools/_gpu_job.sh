cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stats_stf.py -m gpu -x -q < /dev/null 2>&1 | grep -E "passed|failed|error" | tail -n 2 > gpurun_out/r03at.txt
timeout 300 python tools/time_stats.py < /dev/null 2>&1 | tail -n 3 | cut -c1-120 >> gpurun_out/r03at.txt
