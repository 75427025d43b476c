#!/bin/bash
# developer job (round 6): the labelling from the tile pass's candidate lists -- parity tests, an interleaved A/B of the bench step
# (developer library: AB_NO_CAND_LISTS=1 = round 5's labelling), a kernel trace of the release library
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1200 python -m pytest tests/test_gpu_detect_affine.py tests/test_golden.py tests/test_gpu_tile_stats.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/r06b_tests.log 2>&1
echo "tests rc=$?" > gpurun_out/r06b_rc.txt; tail -5 gpurun_out/r06b_tests.log
DEV=$ROOT/astroburst_amd/libastroburst_hip_dev.so
if [ -f $DEV ]; then
  for i in 1 2 3; do
    AB_LIB_PATH=$DEV timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r06b_ab_cand_$i.json 2> /dev/null
    AB_LIB_PATH=$DEV AB_NO_CAND_LISTS=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r06b_ab_nocand_$i.json 2> /dev/null
  done
fi
if [ -f $DEV ]; then   # f2: where scms_kernel's time goes: the developer library's stage cuts, same box, same lights
  ( echo "# tools/time_batch.py on 64 x 4096^2 lights, developer library: AB_BATCH_STAGE 1 = no sort, 2 = no LDS copy of the sorted samples, 3 = no epilogue loop (each with max_iterations = 0)";
    echo "== full kernel"; AB_LIB_PATH=$DEV timeout 300 python tools/time_batch.py 2>&1 | grep -v amdgpu.ids;
    for st in 1 2 3; do echo "== AB_BATCH_STAGE=$st"; AB_LIB_PATH=$DEV AB_BATCH_STAGE=$st timeout 300 python tools/time_batch.py 2>&1 | grep -v amdgpu.ids; done ) > gpurun_out/r06b_batch_stage_cuts.txt
fi
timeout 600 python bench.py > gpurun_out/r06b_bench.json 2> gpurun_out/r06b_bench.err; echo "bench rc=$?" >> gpurun_out/r06b_rc.txt
OUT=$ROOT/gpurun_out/prof_r06b; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.log 2>&1)
DB=$(ls $OUT/*.db 2>/dev/null | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/r06b_kernel_stats.txt && rm -f $OUT/*.db
cat gpurun_out/r06b_rc.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06b_ab_*.json")) + ["gpurun_out/r06b_bench.json"]:
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(f, d["ms_per_step"], d["config"]["stage_ms"], d["config"].get("fallbacks"))
    except Exception as e:
        print(f, "ERR", e)
PY
head -30 gpurun_out/r06b_kernel_stats.txt | cut -c1-200
