#!/bin/bash
mkdir -p gpurun_out
run() { env "$@" timeout 900 python bench.py --config C3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$*', d['ms_per_step'], d['config']['stage_ms']['register_15_frames_estimate_and_warp'])" >> gpurun_out/r05p_c3_sweep.txt; }
for rep in 1 2; do
run AB_NOOP=1
run AB_REGISTER_GROUP=1
run AB_REGISTER_GROUP=2
run AB_REGISTER_GROUP=1 AB_TILE_CHUNK=2 AB_TILE_FIRST=2
run AB_REGISTER_GROUP=2 AB_TILE_CHUNK=2 AB_TILE_FIRST=3
run AB_REGISTER_GROUP=1 AB_TILE_CHUNK=1 AB_TILE_FIRST=2
run AB_REGISTER_GROUP=2 AB_TILE_CHUNK=4 AB_TILE_FIRST=3
done
cat gpurun_out/r05p_c3_sweep.txt
