#!/bin/bash
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
VAR=$PWD/astroburst_amd/csrc/build/variants/libab_w4nofb.so
( echo "# time_warp.py: warp4 as built / without its per-pixel fallback (timing only) / round 5's kernel"
  AB_LIB_PATH=$DEV timeout 300 python tools/time_warp.py 2>&1 | grep -v amdgpu | head -4 | cut -c1-60
  AB_LIB_PATH=$VAR timeout 300 python tools/time_warp.py 2>&1 | grep -v amdgpu | head -4 | cut -c1-60
  AB_LIB_PATH=$DEV AB_WARP_TWO=1 timeout 300 python tools/time_warp.py 2>&1 | grep -v amdgpu | head -4 | cut -c1-60 ) > gpurun_out/r06_warp4_probe.txt 2>&1
cat gpurun_out/r06_warp4_probe.txt
