#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 900 python bench.py --config C3 > gpurun_out/r06z_bench_C3_run$i.json 2> gpurun_out/r06z_bench_C3_run$i.err
  python -c "
import json,sys; d=json.loads(open('gpurun_out/r06z_bench_C3_run$i.json').read().strip().split('\n')[-1]); print($i, d['ms_per_step'], d['config']['stage_ms'])"
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
