cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_stack.py tests/test_gpu_batch.py tests/test_gpu_fits.py -m gpu -x -q < /dev/null > gpurun_out/r04r_pytest.log 2>&1
tail -n 4 gpurun_out/r04r_pytest.log
N_LIST=100,128,200,256,320,512 timeout 600 python tools/time_stack_deep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04r_deep.txt
timeout 600 python tools/time_median_combine.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04r_median.txt
