#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_detect_affine.py tests/test_gpu_tile_stats.py tests/test_gpu_full_size.py -m gpu -x -q --timeout=900 --timeout-method=thread -p no:cacheprovider > gpurun_out/r05ak_tests.txt 2>&1; tail -6 gpurun_out/r05ak_tests.txt
timeout 900 python bench.py --config C3 --no-cpu-baseline > gpurun_out/r05ak_bench_C3.json 2> gpurun_out/r05ak_c3.err; python -c "
import json;d=json.loads(open('gpurun_out/r05ak_bench_C3.json').read().strip().split('\n')[-1]);print(d['ms_per_step'], d['config']['stage_ms'])"
REPS=10 timeout 200 python tools/time_register.py 2>&1 | grep -v "^/opt" | cut -c1-120
