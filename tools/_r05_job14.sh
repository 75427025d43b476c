#!/bin/bash
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PT="python -m pytest -m gpu -x -q --timeout=600 --timeout-method=thread -p no:cacheprovider"
timeout 900 $PT tests/test_gpu_detect_affine.py tests/test_gpu_full_size.py > gpurun_out/r05o_tests.txt 2>&1; echo "tests rc=$?" > gpurun_out/r05o_rc.txt
for rep in 1 2; do
  REPS=10 timeout 300 python tools/time_register.py >> gpurun_out/r05o_register.txt 2>&1
  timeout 900 python bench.py --config C3 --no-cpu-baseline > gpurun_out/r05o_C3_$rep.json 2>> gpurun_out/r05o_C3.err
done
cat gpurun_out/r05o_rc.txt; tail -3 gpurun_out/r05o_tests.txt
grep -v "^/opt" gpurun_out/r05o_register.txt | cut -c1-200
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05o_C3_*.json")):
    d = json.loads(open(f).read().strip().split("\n")[-1]); print(f, d["ms_per_step"], d["config"].get("stage_ms"))
PY
