#!/bin/bash
# developer job (round 6): the registration call's timeline with and without its warps (developer library: AB_ABLATE_WARP=1), the host's view
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export AB_LIB_PATH=$ROOT/astroburst_amd/libastroburst_hip_dev.so
for rep in 1 2; do
  REPS=10 timeout 300 python tools/time_register.py >> gpurun_out/r06r_ablate.txt 2>&1
  REPS=10 AB_ABLATE_WARP=1 timeout 300 python tools/time_register.py >> gpurun_out/r06r_ablate.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
for v in full nowarp; do
  OUT=$ROOT/gpurun_out/prof_r06r_$v; mkdir -p $OUT
  if [ $v = nowarp ]; then export AB_ABLATE_WARP=1; else unset AB_ABLATE_WARP; fi
  REPS=6 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/tools/time_register.py > $OUT/log.txt 2>&1
  python $ROOT/tools/register_timeline.py $(ls $OUT/*results.db | head -1) > $ROOT/gpurun_out/r06r_timeline_$v.txt 2>&1
  rm -f $OUT/*.db
done
cd $ROOT
grep -v "^/opt" gpurun_out/r06r_ablate.txt | cut -c1-180
grep "^call" gpurun_out/r06r_timeline_full.txt | head -9; grep "^call" gpurun_out/r06r_timeline_nowarp.txt | head -9
unset AB_ABLATE_WARP
REPS=3 AB_UPLOAD_TRACE=1 timeout 200 python tools/time_register.py > gpurun_out/r06r_tl.tmp 2>&1
awk '/ab timeline/ {print}' gpurun_out/r06r_tl.tmp | tail -200 > gpurun_out/r06r_host_timeline.txt
sed -n 1,60p gpurun_out/r06r_timeline_full.txt | cut -c1-160
