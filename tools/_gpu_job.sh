set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -n 8 > gpurun_out/r03h_pytest_all.log
python bench.py --no-cpu-baseline > gpurun_out/r03h_bench.json 2> gpurun_out/r03h_bench.err
python tools/time_register.py > gpurun_out/r03h_time_register.txt 2>&1
