#!/bin/bash
# developer job: the batched phase correlation of stack_images(align) -- tests, then C1 batched against pair by pair
mkdir -p gpurun_out
PT="python -m pytest -m gpu -x -q --timeout=600 --timeout-method=thread -p no:cacheprovider"
timeout 900 $PT tests/test_gpu_phasecorr.py tests/test_gpu_stack.py tests/test_gpu_full_size.py -k "phase or align or c1 or stack_images or correl" > gpurun_out/r05k_tests.txt 2>&1; echo "tests rc=$?" >> gpurun_out/r05k_rc.txt
timeout 600 $PT tests/test_gpu_concurrent.py > gpurun_out/r05k_concurrent.txt 2>&1; echo "concurrent rc=$?" >> gpurun_out/r05k_rc.txt
for rep in 1 2; do
  timeout 600 python bench.py --config C1 --no-cpu-baseline > gpurun_out/r05k_C1_batched_$rep.json 2> gpurun_out/r05k_C1.err
  AB_STACK_PAIRWISE=1 timeout 600 python bench.py --config C1 --no-cpu-baseline > gpurun_out/r05k_C1_pairwise_$rep.json 2>> gpurun_out/r05k_C1.err
done
cat gpurun_out/r05k_rc.txt; tail -3 gpurun_out/r05k_tests.txt; tail -3 gpurun_out/r05k_concurrent.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05k_C1_*.json")):
    try:
        d = json.loads(open(f).read().strip().split("\n")[-1])
        print(f, d["ms_per_step"], d["config"].get("stage_ms"))
    except Exception as e:
        print(f, "ERR", e)
PY
tail -5 gpurun_out/r05k_C1.err
