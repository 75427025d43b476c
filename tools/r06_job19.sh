#!/bin/bash
# developer job (round 6): when the warps of a registration call start; C1's kernels
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r06s; mkdir -p $OUT
REPS=6 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/tools/time_register.py > $OUT/log.txt 2>&1
python $ROOT/tools/register_timeline.py $(ls $OUT/*results.db | head -1) > $ROOT/gpurun_out/r06s_timeline.txt 2>&1
rm -f $OUT/*.db
OUT=$ROOT/gpurun_out/prof_r06s_c1; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/bench.py --config C1 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/log.txt 2>&1
python $ROOT/tools/rocpd_summary.py $(ls $OUT/*results.db | head -1) > $ROOT/gpurun_out/r06s_c1_kernels.txt 2>&1
rm -f $OUT/*.db
cd $ROOT
grep -E "^call|warp k starts|first start" gpurun_out/r06s_timeline.txt | cut -c1-400
head -40 gpurun_out/r06s_c1_kernels.txt | cut -c1-180
