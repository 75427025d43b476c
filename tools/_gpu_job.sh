cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q < /dev/null 2>&1 | grep -E "passed|failed|error" | tail -n 3 > gpurun_out/r03w_pytest_gpu.log
