#!/bin/bash
# developer job (round 6, first call): the release library through the whole GPU suite, the developer library through the tests that
# need its switches, the default bench line
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06a_gpu_tests.log 2>&1; echo "gpu tests rc=$?" > gpurun_out/r06a_rc.txt
tail -3 gpurun_out/r06a_gpu_tests.log
if [ -f astroburst_amd/libastroburst_hip_dev.so ]; then
  AB_LIB_PATH=$PWD/astroburst_amd/libastroburst_hip_dev.so timeout 900 python -m pytest tests -m gpu -q \
     -k "superseded or fed_pipeline or chained_detection or round1_kernel or resident_abort or resident_timeout or crowded_tiles" > gpurun_out/r06a_dev_tests.log 2>&1
  echo "dev tests rc=$?" >> gpurun_out/r06a_rc.txt
  tail -3 gpurun_out/r06a_dev_tests.log
fi
timeout 600 python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err; echo "bench rc=$?" >> gpurun_out/r06a_rc.txt
cat gpurun_out/r06a_rc.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06a_bench.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("frac_in_step"), d["roofline"].get("frac_profile"), d["config"]["stage_ms"], d["config"]["fallbacks"])
PY
