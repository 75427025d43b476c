#!/bin/bash
mkdir -p gpurun_out
for rep in 1 2; do
  for w in 3 4 6 8 12; do
    REPS=10 AB_REGISTER_WORKERS=$w timeout 300 python tools/time_register.py >> gpurun_out/r05j_workers.txt 2>&1
  done
  REPS=10 AB_NO_WARP_STREAM=1 timeout 300 python tools/time_register.py >> gpurun_out/r05j_workers.txt 2>&1
  REPS=10 AB_REGISTER_GROUP=2 AB_REGISTER_WORKERS=6 timeout 300 python tools/time_register.py >> gpurun_out/r05j_workers.txt 2>&1
  REPS=10 AB_REGISTER_GROUP=8 AB_REGISTER_WORKERS=4 timeout 300 python tools/time_register.py >> gpurun_out/r05j_workers.txt 2>&1
done
grep -v "^/opt" gpurun_out/r05j_workers.txt | cut -c1-160
