set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_batch.py tests/test_gpu_stack.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -n 5 > gpurun_out/r03k_pytest.log
python tools/time_batch.py > gpurun_out/r03k_time_batch.txt 2>&1
python tools/chunk_check.py 2>&1 | tail -n 3 > gpurun_out/r03k_chunk_check.txt
python bench.py --config C1 > gpurun_out/r03k_bench_C1.json 2> gpurun_out/r03k_bench_C1.err
bash tools/pmc_stack_sq.sh > gpurun_out/r03k_sq.txt 2>&1
