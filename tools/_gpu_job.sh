cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_detect_affine.py tests/test_gpu_full_size.py -m gpu -x -q < /dev/null 2>&1 | head -n 3 > gpurun_out/r03aj_pytest.log
timeout 600 python bench.py --config C3 --no-cpu-baseline < /dev/null 2>/dev/null | tail -n 1 > gpurun_out/r03aj_C3.json
for i in 1 2 3; do timeout 120 python tools/time_register.py < /dev/null 2>&1 | tail -n 1 | cut -c1-70 >> gpurun_out/r03aj_pytest.log; done
