#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel-trace stats + the two PMC passes of the default bench
# command, each in its own rocprofv3 run (PMC is never combined with other trace domains).
# Output: gpurun_out/prof_$TAG/{trace,fetch,write}_results.db ; summarise with tools/summarise_profiles.py
TAG=${1:-r01}
STEPS=${2:-5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline > $OUT/bench_trace.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT -o fetch -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT -o write -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_write.log 2>&1
# summarise on the box (the raw sqlite traces are tens of MB each) and keep only the text
cd $ROOT && python tools/summarise_profiles.py $TAG $OUT && rm -f $OUT/*.db
ls -la $OUT
