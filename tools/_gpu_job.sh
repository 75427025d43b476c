set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_stack.py tests/test_golden.py -m gpu -q 2>&1 | tail -n 6 > gpurun_out/r03f_pytest.log
bash tools/profile_bench.sh r03f 5 > gpurun_out/r03f_profile.log 2>&1
python bench.py --no-cpu-baseline > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err
