#!/bin/bash
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PT="python -m pytest -m gpu -x -v --timeout=600 --timeout-method=thread -p no:cacheprovider"
timeout 900 $PT tests/test_gpu_detect_affine.py -k "grouped or brightest or register_frames or align_pairs" > gpurun_out/r05f_detect.txt 2>&1; echo "detect rc=$?" >> gpurun_out/r05f_rc.txt
REPS=3 AB_TRACE=1 timeout 300 python tools/time_register.py 2>&1 | grep -c "redone in full" > gpurun_out/r05f_fallbacks.txt
for rep in 1 2 3; do
  REPS=10 timeout 300 python tools/time_register.py >> gpurun_out/r05f_register_ab.txt 2>&1
  REPS=10 AB_LABEL_LEGACY=1 timeout 300 python tools/time_register.py >> gpurun_out/r05f_register_ab.txt 2>&1
done
grep -v "^/opt" gpurun_out/r05f_register_ab.txt | cut -c1-200
echo "fallback lines in 5 calls: $(cat gpurun_out/r05f_fallbacks.txt)"
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r05f; mkdir -p $OUT
REPS=6 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/tools/time_register.py > $OUT/log.txt 2>&1
python $ROOT/tools/rocpd_summary.py $(ls $OUT/*results.db | head -1) > $ROOT/gpurun_out/r05f_kernels.txt 2>&1
rm -f $OUT/*.db
cd $ROOT
head -20 gpurun_out/r05f_kernels.txt | cut -c1-200
cat gpurun_out/r05f_rc.txt; tail -3 gpurun_out/r05f_detect.txt
