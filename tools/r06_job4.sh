#!/bin/bash
# developer job (round 6): the whole GPU suite on the release library; A/B of the helper-thread fits and of stream priorities (developer
# library); the tile kernel's phase cycles
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06d_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r06d_rc.txt; tail -4 gpurun_out/r06d_tests.log
DEV=$ROOT/astroburst_amd/libastroburst_hip_dev.so
run() { tag=$1; shift; env AB_LIB_PATH=$DEV "$@" timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(sys.argv[1], d["ms_per_step"], d["config"]["stage_ms"]["register_63_frames_estimate_and_warp"])' "$tag"; }
( for i in 1 2 3; do
    run default AB_X=0
    run no_fit_threads AB_NO_FIT_THREADS=1
    run tile_prio_high AB_TILE_PRIO=-1
    run warp_prio_low AB_WARP_PRIO=1
    run tile_high_warp_low AB_TILE_PRIO=-1 AB_WARP_PRIO=1
    run workers16 AB_REGISTER_WORKERS=16
    run chunk4 AB_TILE_CHUNK=4
  done ) > gpurun_out/r06d_ab.txt 2>&1
cat gpurun_out/r06d_ab.txt
( echo "# build/tile_stream_bench_timing (AB_TILE_TIMING): phase cycles of tile_background_stream's tiles, round 6 (pass 2 in batches of sixteen)"; for m in 2; do build/tile_stream_bench_timing $m 1; build/tile_stream_bench_timing $m 8; done; echo "# without the timing instrumentation"; build/tile_stream_bench 2 1; build/tile_stream_bench 2 8 ) > gpurun_out/r06d_tile_phases.txt 2>&1
head -30 gpurun_out/r06d_tile_phases.txt | cut -c1-220
cat gpurun_out/r06d_rc.txt
