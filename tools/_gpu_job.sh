cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_stack.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -n 4 > gpurun_out/r03n_pytest.log
V=astroburst_amd/csrc/build/variants
python tools/ab_stack_variants.py --rounds 3 --clean default $V/libab_prev.so $V/libab_nofull0.so $V/libab_ieeesqrt.so $V/libab_defer3.so 2>&1 | grep -v "Warning\|frames = \|amdgpu.ids" > gpurun_out/r03n_variants.txt
