#!/bin/bash
# round 6: the committed tree once more after the removed warp experiment: smoke, the resampling / registration tests, the bench line
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -x -q -k "warp or resample or affine or rowband or shift or align or golden" 2>&1 | tail -2
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/r06last2_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r06last2_bench.json')); print(d['ms_per_step'], d['roofline']['frac'], d['config']['stage_ms'])"
