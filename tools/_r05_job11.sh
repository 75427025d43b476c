#!/bin/bash
# developer job: the whole GPU suite, the bench line, the rocprofv3 summaries of the bench command
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=900 --timeout-method=thread -p no:cacheprovider > gpurun_out/r05l_suite.txt 2>&1; echo "suite rc=$?" > gpurun_out/r05l_rc.txt
timeout 600 python bench.py > gpurun_out/r05l_bench.json 2> gpurun_out/r05l_bench.err; echo "bench rc=$?" >> gpurun_out/r05l_rc.txt
timeout 1800 bash tools/profile_bench.sh r05l 5 > gpurun_out/r05l_profile.log 2>&1; echo "profile rc=$?" >> gpurun_out/r05l_rc.txt
cat gpurun_out/r05l_rc.txt; tail -4 gpurun_out/r05l_suite.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05l_bench.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], d["value"], d["roofline"], d["config"]["stage_ms"])
PY
ls gpurun_out/prof_r05l profiles | tail -20
