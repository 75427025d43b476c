set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_stack.py tests/test_golden.py tests/test_gpu_fits.py tests/test_gpu_full_size.py tests/test_gpu_sharded.py tests/test_gpu_batch.py -m gpu -q 2>&1 | tail -n 15 > gpurun_out/r03d_pytest_default.log
V=astroburst_amd/csrc/build/variants
python tools/ab_stack_variants.py --rounds 2 --clean default $V/libab_rawalways.so 2>&1 | grep -v "Warning\|frames = \|amdgpu.ids" > gpurun_out/r03d_variants.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -n 15 > gpurun_out/r03d_pytest_all.log
