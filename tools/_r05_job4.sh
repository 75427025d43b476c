#!/bin/bash
# developer job: region-split tile labelling + the two-pass selection against the two-pass labelling; the warp's occupancy cap
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PT="python -m pytest -m gpu -x -v --timeout=600 --timeout-method=thread -p no:cacheprovider"
timeout 900 $PT tests/test_gpu_detect_affine.py -k "grouped or brightest or register_frames or align_pairs" > gpurun_out/r05e_detect.txt 2>&1; echo "detect rc=$?" >> gpurun_out/r05e_rc.txt
for rep in 1 2; do
  REPS=10 timeout 300 python tools/time_register.py >> gpurun_out/r05e_register_ab.txt 2>&1
  REPS=10 AB_LABEL_LEGACY=1 timeout 300 python tools/time_register.py >> gpurun_out/r05e_register_ab.txt 2>&1
  for kb in 24 40 64; do
    REPS=10 AB_LABEL_LEGACY=1 AB_WARP_LDS_KB=$kb timeout 300 python tools/time_register.py >> gpurun_out/r05e_register_ab.txt 2>&1
  done
done
grep -v "^/opt" gpurun_out/r05e_register_ab.txt | cut -c1-200
cd /tmp && export TMPDIR=/tmp
for v in tiled legacy; do
  OUT=$ROOT/gpurun_out/prof_r05e_$v; mkdir -p $OUT
  if [ $v = legacy ]; then export AB_LABEL_LEGACY=1; else unset AB_LABEL_LEGACY; fi
  REPS=6 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/tools/time_register.py > $OUT/log.txt 2>&1
  python $ROOT/tools/rocpd_summary.py $(ls $OUT/*results.db | head -1) > $ROOT/gpurun_out/r05e_kernels_$v.txt 2>&1
  rm -f $OUT/*.db
done
unset AB_LABEL_LEGACY
cd $ROOT
head -16 gpurun_out/r05e_kernels_tiled.txt | cut -c1-200
head -16 gpurun_out/r05e_kernels_legacy.txt | cut -c1-200
timeout 600 $PT tests/test_gpu_multirank.py -k "bands" > gpurun_out/r05e_multirank.txt 2>&1; echo "multirank rc=$?" >> gpurun_out/r05e_rc.txt
cat gpurun_out/r05e_rc.txt
tail -5 gpurun_out/r05e_detect.txt; grep -n "PASSED\|FAILED\|hung" gpurun_out/r05e_multirank.txt | tail -10
