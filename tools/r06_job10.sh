#!/bin/bash
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
run() { tag=$1; shift; env AB_LIB_PATH=$DEV "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], d["ms_per_step"], d["config"]["stage_ms"]["register_63_frames_estimate_and_warp"])' "$tag"; }
( echo "# bench.py --steps 20, developer library, interleaved on one box: ms per step, registration ms (default: warp stream low, tile stream normal)"
  for i in 1 2 3; do
    run default AB_X=0
    run tile_low AB_TILE_PRIO=1
    run tile_low_chunk16 AB_TILE_PRIO=1 AB_TILE_CHUNK=16
    run chunk16 AB_TILE_CHUNK=16
    run workers8 AB_REGISTER_WORKERS=8
    run tile_first8 AB_TILE_FIRST=0
  done ) > gpurun_out/r06j_ab.txt 2>&1
cat gpurun_out/r06j_ab.txt
