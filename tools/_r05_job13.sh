#!/bin/bash
# developer job: the registration call's timeline (round 5) and what C3 is made of
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r05n; mkdir -p $OUT
REPS=6 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/tools/time_register.py > $OUT/log.txt 2>&1
python $ROOT/tools/register_timeline.py $(ls $OUT/*results.db | head -1) > $ROOT/gpurun_out/r05n_register_timeline.txt 2>&1
rm -f $OUT/*.db
OUT=$ROOT/gpurun_out/prof_r05n_c3; mkdir -p $OUT
AB_TRACE=1 timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/bench.py --config C3 --no-cpu-baseline > $OUT/log.txt 2> $OUT/err.txt
python $ROOT/tools/rocpd_summary.py $(ls $OUT/*results.db | head -1) > $ROOT/gpurun_out/r05n_c3_kernels.txt 2>&1
python $ROOT/tools/register_timeline.py $(ls $OUT/*results.db | head -1) > $ROOT/gpurun_out/r05n_c3_timeline.txt 2>&1
rm -f $OUT/*.db
cd $ROOT
grep -c "redone in full" gpurun_out/prof_r05n_c3/err.txt
head -60 gpurun_out/r05n_register_timeline.txt | cut -c1-200
head -24 gpurun_out/r05n_c3_kernels.txt | cut -c1-200
head -50 gpurun_out/r05n_c3_timeline.txt | cut -c1-200
