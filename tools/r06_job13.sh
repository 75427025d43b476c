#!/bin/bash
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
run() { tag=$1; shift; env AB_LIB_PATH=$DEV "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], d["ms_per_step"], d["config"]["stage_ms"]["register_63_frames_estimate_and_warp"])' "$tag"; }
( echo "# bench.py --steps 20, developer library, interleaved on one box: ms per step, registration ms (default: workers normal, tile stream normal, warps low)"
  for i in 1 2 3; do
    run default AB_X=0
    run workers_high AB_WORKER_PRIO=-1
    run workers_high_tile_low AB_WORKER_PRIO=-1 AB_TILE_PRIO=1
    run workers_high_warp_normal AB_WORKER_PRIO=-1 AB_WARP_PRIO=0
    run tile_low_warp_low AB_TILE_PRIO=1
    run hwq8 GPU_MAX_HW_QUEUES=8
    run hwq8_workers_high GPU_MAX_HW_QUEUES=8 AB_WORKER_PRIO=-1
  done ) > gpurun_out/r06m_ab.txt 2>&1
cat gpurun_out/r06m_ab.txt
