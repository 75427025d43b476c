#!/bin/bash
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PT="python -m pytest -m gpu -x -v --timeout=600 --timeout-method=thread -p no:cacheprovider"
timeout 600 $PT tests/test_gpu_detect_affine.py -k "grouped or brightest or register_frames" > gpurun_out/r05i_detect.txt 2>&1; echo "detect rc=$?" >> gpurun_out/r05i_rc.txt
for rep in 1 2; do
  for wg in 512 1024 2048 4096; do
    REPS=10 AB_LABEL_TILE_WG=$wg timeout 300 python tools/time_register.py >> gpurun_out/r05i_register.txt 2>&1
  done
done
cd /tmp && export TMPDIR=/tmp
for wg in 1024 2048; do
OUT=$ROOT/gpurun_out/prof_r05i_$wg; mkdir -p $OUT
REPS=6 AB_LABEL_TILE_WG=$wg timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/tools/time_register.py > $OUT/log.txt 2>&1
python $ROOT/tools/rocpd_summary.py $(ls $OUT/*results.db | head -1) > $ROOT/gpurun_out/r05i_kernels_$wg.txt 2>&1
rm -f $OUT/*.db
done
cd $ROOT
cat gpurun_out/r05i_rc.txt; tail -2 gpurun_out/r05i_detect.txt
grep -v "^/opt" gpurun_out/r05i_register.txt | cut -c1-200
grep "label_tile\|warp_kernel\|tile_background_stream" gpurun_out/r05i_kernels_*.txt | cut -c1-200
