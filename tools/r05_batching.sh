#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/r05as_group.txt; : > $out
for rep in 1 2; do
  for cfg in "AB_NOOP=1" "AB_REGISTER_GROUP=5" "AB_REGISTER_GROUP=6" "AB_REGISTER_GROUP=6 AB_TILE_CHUNK=6" "AB_REGISTER_GROUP=3 AB_REGISTER_WORKERS=21" "AB_REGISTER_GROUP=8 AB_TILE_CHUNK=8"; do
    env $cfg REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" >> $out
  done
done
cut -c1-150 $out
