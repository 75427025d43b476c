#!/bin/bash
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
VAR=$PWD/astroburst_amd/csrc/build/variants/libab_rot.so
AB_LIB_PATH=$VAR timeout 600 python -m pytest tests/test_gpu_tile_stats.py tests/test_gpu_detect_affine.py -m gpu -x -q 2>&1 | tail -2
run() { tag=$1; lib=$2; shift 2; env AB_LIB_PATH=$lib "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], d["ms_per_step"], d["config"]["stage_ms"]["register_63_frames_estimate_and_warp"])' "$tag"; }
( echo "# rounds on wave (tile mod 4) instead of wave 0: bench.py --steps 20, developer builds, interleaved"
  for i in 1 2 3 4; do run wave0 $DEV AB_X=0; run rotated $VAR AB_X=0; done ) > gpurun_out/r06n_ab.txt 2>&1
cat gpurun_out/r06n_ab.txt
