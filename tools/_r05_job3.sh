#!/bin/bash
# developer job: the chained detection against the mid-join form, kernel traces of the tiled and the two-pass labelling, tests
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PT="python -m pytest -m gpu -x -v --timeout=600 --timeout-method=thread -p no:cacheprovider"
for rep in 1 2; do
  REPS=10 timeout 300 python tools/time_register.py >> gpurun_out/r05d_register_ab.txt 2>&1
  REPS=10 AB_LABEL_LEGACY=1 timeout 300 python tools/time_register.py >> gpurun_out/r05d_register_ab.txt 2>&1
  REPS=10 AB_LABEL_LEGACY=1 AB_DETECT_MIDJOIN=1 timeout 300 python tools/time_register.py >> gpurun_out/r05d_register_ab.txt 2>&1
  REPS=10 AB_DETECT_MIDJOIN=1 timeout 300 python tools/time_register.py >> gpurun_out/r05d_register_ab.txt 2>&1
done
grep -v "^/opt" gpurun_out/r05d_register_ab.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for v in tiled legacy; do
  OUT=$ROOT/gpurun_out/prof_r05d_$v; mkdir -p $OUT
  if [ $v = legacy ]; then export AB_LABEL_LEGACY=1; else unset AB_LABEL_LEGACY; fi
  REPS=6 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/tools/time_register.py > $OUT/log.txt 2>&1
  python $ROOT/tools/rocpd_summary.py $(ls $OUT/*results.db | head -1) > $ROOT/gpurun_out/r05d_kernels_$v.txt 2>&1
  rm -f $OUT/*.db
done
unset AB_LABEL_LEGACY
cd $ROOT
head -24 gpurun_out/r05d_kernels_tiled.txt | cut -c1-200
head -24 gpurun_out/r05d_kernels_legacy.txt | cut -c1-200
timeout 900 $PT tests/test_gpu_detect_affine.py -k "grouped or brightest or register_frames or align_pairs" > gpurun_out/r05d_detect.txt 2>&1; echo "detect rc=$?" >> gpurun_out/r05d_rc.txt
timeout 300 $PT tests/test_gpu_masked.py -k "large_iteration" > gpurun_out/r05d_masked.txt 2>&1; echo "masked rc=$?" >> gpurun_out/r05d_rc.txt
timeout 600 $PT tests/test_gpu_multirank.py -k "bands" > gpurun_out/r05d_multirank.txt 2>&1; echo "multirank rc=$?" >> gpurun_out/r05d_rc.txt
cat gpurun_out/r05d_rc.txt
tail -5 gpurun_out/r05d_detect.txt; tail -5 gpurun_out/r05d_masked.txt; grep -n "bands rank\|PASSED\|FAILED\|hung" gpurun_out/r05d_multirank.txt | tail -40
