#!/bin/bash
# developer job (round 6): pass 2 of the tile kernel in batches of sixteen + the sharded stack in overlapped row chunks
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests/test_gpu_detect_affine.py tests/test_golden.py tests/test_gpu_tile_stats.py tests/test_gpu_sharded.py tests/test_gpu_multirank.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/r06c_tests.log 2>&1
echo "tests rc=$?" > gpurun_out/r06c_rc.txt; tail -5 gpurun_out/r06c_tests.log
DEV=$ROOT/astroburst_amd/libastroburst_hip_dev.so
for i in 1 2 3; do
  AB_LIB_PATH=$DEV timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r06c_ab_cand_$i.json 2> /dev/null
  AB_LIB_PATH=$DEV AB_NO_CAND_LISTS=1 timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r06c_ab_nocand_$i.json 2> /dev/null
done
timeout 600 python bench.py --force-sharded --no-cpu-baseline > gpurun_out/r06c_bench_sharded.json 2> gpurun_out/r06c_sharded.err; echo "sharded rc=$?" >> gpurun_out/r06c_rc.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err; echo "bench rc=$?" >> gpurun_out/r06c_rc.txt
OUT=$ROOT/gpurun_out/prof_r06c; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_trace.log 2>&1)
DB=$(ls $OUT/*.db 2>/dev/null | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB > gpurun_out/r06c_kernel_stats.txt && rm -f $OUT/*.db
cat gpurun_out/r06c_rc.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06c_ab_*.json")) + ["gpurun_out/r06c_bench.json", "gpurun_out/r06c_bench_sharded.json"]:
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(f, d["ms_per_step"], d["roofline"]["frac"], d["config"]["stage_ms"])
    except Exception as e:
        print(f, "ERR", e)
PY
grep -E "tile_background|label_bgtile|warp_kernel|stack_sigma|comp_select|tri_vote" gpurun_out/r06c_kernel_stats.txt | cut -c1-200
