cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q < /dev/null 2>&1 | grep -E "passed|failed|error" | tail -n 3 > gpurun_out/r03x_pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r03x_bench.json 2> gpurun_out/r03x_bench.err < /dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03x_smoke.txt 2>&1 < /dev/null
