#!/bin/bash
# round 6: C2 (the bench step) with the warp kernel's pieces in XCD bands (default) against row-major ids (variant), interleaved
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
VAR=$PWD/astroburst_amd/csrc/build/variants/libab_rowmajor.so
run() { tag=$1; lib=$2; AB_LIB_PATH=$lib timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); s=d["config"]["stage_ms"]; print(sys.argv[1], d["ms_per_step"], s["register_63_frames_estimate_and_warp"], d["roofline"]["frac"])' "$tag"; }
( echo "# bench.py (C2), developer builds, interleaved: XCD bands (default) vs row-major piece ids (-DAB_WARP_ROW_MAJOR)"
  for i in 1 2 3 4; do run bands $DEV; run rowmajor $VAR; done ) > gpurun_out/r06w_ab_c2.txt 2>&1
cat gpurun_out/r06w_ab_c2.txt
