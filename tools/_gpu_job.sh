cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 120 build/tile_bench 0 > gpurun_out/r03o_tile_bench.txt 2>&1
timeout 120 build/tile_bench 1 >> gpurun_out/r03o_tile_bench.txt 2>&1
AB_TILE_MAP=1 timeout 900 python -m pytest tests/test_gpu_tile_stats.py -m gpu -x -q 2>&1 | tail -n 15 > gpurun_out/r03o_pytest_tiles.log
