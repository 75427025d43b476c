#!/bin/bash
# developer job (round 6): full suite with the low-priority warp stream as the default; A/B of that default; the LDS-exclusion hypothesis
# (three tile workgroups per CU instead of four, with the labelling kernel at 36 KB instead of 62)
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06e_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r06e_rc.txt; tail -4 gpurun_out/r06e_tests.log
DEV=$ROOT/astroburst_amd/libastroburst_hip_dev.so
VAR=$ROOT/astroburst_amd/csrc/build/variants/libab_bgsmall.so
run() { tag=$1; lib=$2; shift 2; env AB_LIB_PATH=$lib "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], d["ms_per_step"], d["config"]["stage_ms"]["register_63_frames_estimate_and_warp"], d["config"]["fallbacks"]["frames_redone"])' "$tag"; }
( for i in 1 2 3; do
    run default_warp_low $DEV AB_X=0
    run warp_normal $DEV AB_WARP_PRIO=0
    run tile_pad4 $DEV AB_TILE_PAD_KB=4
    run bgsmall $VAR AB_X=0
    run bgsmall_tile_pad4 $VAR AB_TILE_PAD_KB=4
    run workers16 $DEV AB_REGISTER_WORKERS=16
    run group8 $DEV AB_REGISTER_GROUP=8
  done ) > gpurun_out/r06e_ab.txt 2>&1
cat gpurun_out/r06e_ab.txt
cat gpurun_out/r06e_rc.txt
