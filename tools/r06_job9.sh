#!/bin/bash
# developer job (round 6): 257 .. 512 frames -- two lanes per pixel (stack_pair.hip, the default) against the LDS-staged wave-per-pixel kernel
# with tree sums (developer library, AB_STACK_NO_PAIR=1)
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
( echo "# 257 .. 512 frames x 4096^2 (tools/time_stack_deep.py, developer library)"
  echo "## default: stack_pair.hip"
  AB_LIB_PATH=$DEV N_LIST=257,320,512 timeout 900 python tools/time_stack_deep.py 2>&1 | grep -v amdgpu
  echo "## AB_STACK_NO_PAIR=1: stack_wide_tile_kernel<8> (32 adjacent pixels through LDS, window MAD, tree sums)"
  AB_LIB_PATH=$DEV AB_STACK_NO_PAIR=1 N_LIST=257,320,512 timeout 900 python tools/time_stack_deep.py 2>&1 | grep -v amdgpu ) > gpurun_out/r06i_pair_vs_tile.txt
cat gpurun_out/r06i_pair_vs_tile.txt
