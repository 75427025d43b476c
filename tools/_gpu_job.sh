cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 15 > gpurun_out/r03s_pytest_gpu.log
timeout 300 python tools/time_stats.py 2>&1 | tail -n 5 > gpurun_out/r03s_time_stats.txt
timeout 600 python bench.py > gpurun_out/r03s_bench.json 2> gpurun_out/r03s_bench.err
