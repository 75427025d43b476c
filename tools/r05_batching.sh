#!/bin/bash
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r05p_bench.json 2> gpurun_out/r05p_bench.err; echo "bench rc=$?"
timeout 1800 bash tools/profile_bench.sh r05final 5 > gpurun_out/r05p_profile.log 2>&1; echo "profile rc=$?"
timeout 600 python bench.py > gpurun_out/r05p_bench_after.json 2> gpurun_out/r05p_bench2.err
python - <<'PY'
import json
for f in ("gpurun_out/r05p_bench.json","gpurun_out/r05p_bench_after.json"):
    d=json.loads(open(f).read().strip().split("\n")[-1]); r=d["roofline"]
    print(f, d["ms_per_step"], d["config"]["stage_ms"], {k:r[k] for k in ("frac","frac_in_step","frac_sustained","profile_avg_ms","avg_kernel_ms","traffic")})
PY
grep "stack_sigma_clip" gpurun_out/prof_r05final/r05final_kernel_stats.txt | cut -c1-60,88-140
