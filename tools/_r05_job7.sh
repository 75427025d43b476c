#!/bin/bash
mkdir -p gpurun_out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
PT="python -m pytest -m gpu -x -v --timeout=600 --timeout-method=thread -p no:cacheprovider"
timeout 600 $PT tests/test_gpu_detect_affine.py -k "grouped or brightest" > gpurun_out/r05h_detect.txt 2>&1; echo "detect rc=$?" >> gpurun_out/r05h_rc.txt
for rep in 1 2; do
  REPS=10 timeout 300 python tools/time_register.py >> gpurun_out/r05h_register.txt 2>&1
  REPS=10 AB_LABEL_LEGACY=1 timeout 300 python tools/time_register.py >> gpurun_out/r05h_register.txt 2>&1
done
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_r05h; mkdir -p $OUT
REPS=6 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $ROOT/tools/time_register.py > $OUT/log.txt 2>&1
python $ROOT/tools/rocpd_summary.py $(ls $OUT/*results.db | head -1) > $ROOT/gpurun_out/r05h_kernels.txt 2>&1
rm -f $OUT/*.db
cd $ROOT
timeout 600 python bench.py > gpurun_out/r05h_bench.json 2> gpurun_out/r05h_bench.err
cat gpurun_out/r05h_rc.txt; tail -2 gpurun_out/r05h_detect.txt
grep -v "^/opt" gpurun_out/r05h_register.txt | cut -c1-200
head -16 gpurun_out/r05h_kernels.txt | cut -c1-200
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05h_bench.json").read().strip().split("\n")[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["config"]["stage_ms"], d["config"].get("registration"))
PY
