#!/bin/bash
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PT="python -m pytest -m gpu -x -v --timeout=600 --timeout-method=thread -p no:cacheprovider"
timeout 900 $PT tests/test_gpu_stack.py -k "deep_stacks_one_wave or deep_median or deep_stack_at_scale" > gpurun_out/r05g_stack.txt 2>&1; echo "stack rc=$?" >> gpurun_out/r05g_rc.txt
N_LIST=129,160,161,192,200,224,256 timeout 600 python tools/time_stack_deep.py > gpurun_out/r05g_deep.txt 2>&1
N_LIST=160,200,256 AB_STACK_NO_CLASSES=1 timeout 600 python tools/time_stack_deep.py > gpurun_out/r05g_deep_noclasses.txt 2>&1
timeout 600 $PT tests/test_gpu_detect_affine.py -k "grouped or brightest" > gpurun_out/r05g_detect.txt 2>&1; echo "detect rc=$?" >> gpurun_out/r05g_rc.txt
for rep in 1 2; do
  REPS=10 timeout 300 python tools/time_register.py >> gpurun_out/r05g_register.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r05g; mkdir -p $OUT
REPS=6 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/tools/time_register.py > $OUT/log.txt 2>&1
python $ROOT/tools/rocpd_summary.py $(ls $OUT/*results.db | head -1) > $ROOT/gpurun_out/r05g_kernels.txt 2>&1
rm -f $OUT/*.db
cd $ROOT
cat gpurun_out/r05g_rc.txt; tail -3 gpurun_out/r05g_stack.txt; tail -2 gpurun_out/r05g_detect.txt
cat gpurun_out/r05g_deep.txt gpurun_out/r05g_deep_noclasses.txt | grep frames
grep -v "^/opt" gpurun_out/r05g_register.txt | cut -c1-200
head -14 gpurun_out/r05g_kernels.txt | cut -c1-200
