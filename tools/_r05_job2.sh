#!/bin/bash
# developer job: registration A/B (round 4's detection forms against round 5's, interleaved) + the bench line + the fixed tests
mkdir -p gpurun_out
PT="python -m pytest -m gpu -x -v --timeout=300 --timeout-method=thread -p no:cacheprovider"
for rep in 1 2; do
  REPS=10 timeout 300 python tools/time_register.py >> gpurun_out/r05c_register_ab.txt 2>&1
  REPS=10 AB_LABEL_LEGACY=1 AB_DETECT_FULL_RECORDS=1 timeout 300 python tools/time_register.py >> gpurun_out/r05c_register_ab.txt 2>&1
  REPS=10 AB_LABEL_LEGACY=1 timeout 300 python tools/time_register.py >> gpurun_out/r05c_register_ab.txt 2>&1
  REPS=10 AB_DETECT_FULL_RECORDS=1 timeout 300 python tools/time_register.py >> gpurun_out/r05c_register_ab.txt 2>&1
done
timeout 600 python bench.py > gpurun_out/r05c_bench.json 2> gpurun_out/r05c_bench.err
timeout 600 $PT tests/test_gpu_multirank.py -k "bands" > gpurun_out/r05c_multirank.txt 2>&1; echo "multirank rc=$?" >> gpurun_out/r05c_rc.txt
timeout 600 $PT tests/test_gpu_concurrent.py > gpurun_out/r05c_concurrent.txt 2>&1; echo "concurrent rc=$?" >> gpurun_out/r05c_rc.txt
cat gpurun_out/r05c_rc.txt
grep -v "^$" gpurun_out/r05c_register_ab.txt | cut -c1-260
tail -3 gpurun_out/r05c_multirank.txt; tail -3 gpurun_out/r05c_concurrent.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05c_bench.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["config"]["stage_ms"], d["config"].get("registration"))
PY
