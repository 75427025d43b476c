cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stats_stf.py -m gpu -x -q 2>&1 | tail -n 15 > gpurun_out/r03t_pytest.log
timeout 600 python bench.py --no-register > gpurun_out/r03t_bench_noreg.json 2> gpurun_out/r03t_bench.err
