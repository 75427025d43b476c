#!/bin/bash
# developer job (round 6): the labelling by sub-tiles (LDS 62 -> 21 KB): parity, then sub-tile heights 32 / 64 / 128 / 256 interleaved
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1200 python -m pytest tests/test_gpu_detect_affine.py tests/test_golden.py tests/test_gpu_full_size.py tests/test_gpu_multirank.py tests/test_gpu_concurrent.py -m gpu -x -q > gpurun_out/r06f_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r06f_rc.txt; tail -4 gpurun_out/r06f_tests.log
DEV=$ROOT/astroburst_amd/libastroburst_hip_dev.so
V=$ROOT/astroburst_amd/csrc/build/variants
run() { tag=$1; lib=$2; shift 2; env AB_LIB_PATH=$lib "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], d["ms_per_step"], d["config"]["stage_ms"]["register_63_frames_estimate_and_warp"], d["config"]["fallbacks"]["frames_redone"], d["config"]["fallbacks"]["label_tiles_dense"])' "$tag"; }
( echo "# bench.py --steps 20, developer builds, interleaved on one box: ms per step, registration ms, frames redone, dense tiles";
  echo "# rowsN = label_bgtile workgroups of N rows x 256 columns (LDS: 32 -> 5.6 KB, 64 -> 10.6, 128 -> 21.0 [the default], 256 -> 38 with 3072 runs / 448 slots)";
  for i in 1 2 3; do
    run rows128_default $DEV AB_X=0
    run rows64 $V/libab_bg64.so AB_X=0
    run rows32 $V/libab_bg32.so AB_X=0
    run rows256 $V/libab_bg256.so AB_X=0
    run rows128_group8 $DEV AB_REGISTER_GROUP=8
    run rows128_nocand $DEV AB_NO_CAND_LISTS=1
  done ) > gpurun_out/r06f_ab.txt 2>&1
cat gpurun_out/r06f_ab.txt; cat gpurun_out/r06f_rc.txt
