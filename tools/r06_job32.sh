#!/bin/bash
# round 6: the last library of the round: the whole GPU suite once more, the bench line
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06last_tests.log 2>&1; echo "all gpu tests rc=$?" > gpurun_out/r06last_rc.txt
tail -3 gpurun_out/r06last_tests.log
timeout 600 python bench.py > gpurun_out/r06last_bench.json 2> gpurun_out/r06last_bench.err; echo "bench rc=$?" >> gpurun_out/r06last_rc.txt
python -c "
import json; d=json.loads(open('gpurun_out/r06last_bench.json').read().strip().split('\n')[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['config']['stage_ms'], d['cpu_baseline']['value'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cat gpurun_out/r06last_rc.txt
