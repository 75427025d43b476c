set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_stack.py tests/test_golden.py tests/test_gpu_fits.py tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -n 6 > gpurun_out/r03b_pytest_default.log
V=astroburst_amd/csrc/build/variants
python tools/ab_stack_variants.py --rounds 3 --clean default $V/libab_nohooks.so $V/libab_rawalways.so $V/libab_fused.so 2>&1 | grep -v "Warning\|frames = \|amdgpu.ids" > gpurun_out/r03b_variants.txt
export AB_VARIANTS_DIR=/dev/shm/ab_variants; [ -d $AB_VARIANTS_DIR ] || export AB_VARIANTS_DIR=/tmp/ab_variants
AB_TRACE=1 python tools/ab_stack_variants.py --child default bench 2>&1 | grep "ab_trace" | sort | uniq -c > gpurun_out/r03b_trace.txt
bash tools/pmc_stack_sq.sh > gpurun_out/r03b_sq.txt 2>&1
