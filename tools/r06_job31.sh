#!/bin/bash
# round 6: comp_moments with its component loop fully unrolled (the arrays in registers, not scratch): bench step, interleaved
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
VAR=$PWD/astroburst_amd/csrc/build/variants/libab_unroll.so
AB_LIB_PATH=$VAR timeout 900 python -m pytest tests/test_gpu_detect_affine.py -m gpu -x -q 2>&1 | tail -2
run() { tag=$1; lib=$2; AB_LIB_PATH=$lib timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); s=d["config"]["stage_ms"]; print(sys.argv[1], d["ms_per_step"], s["register_63_frames_estimate_and_warp"])' "$tag"; }
( echo "# bench.py (C2), developer builds, interleaved: detect.hip as built (comp_moments' arrays in scratch) vs -mllvm -pragma-unroll-threshold=1000000"
  for i in 1 2 3 4; do run asbuilt $DEV; run unrolled $VAR; done ) > gpurun_out/r06_moments_ab.txt 2>&1
cat gpurun_out/r06_moments_ab.txt
