#!/bin/bash
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
REL=$PWD/astroburst_amd/libastroburst_hip.so
run() { tag=$1; cfg="$2"; lib=$3; AB_LIB_PATH=$lib timeout 600 python bench.py $cfg --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); s=d["config"]["stage_ms"]; print(sys.argv[1], sys.argv[2], d["ms_per_step"], [v for k,v in s.items() if k.startswith("register")], d["steps"], d["config"].get("fallbacks"))' "$tag" "$cfg"; }
( for i in 1 2; do run release "--config C3" $REL; run dev "--config C3" $DEV; done
  run release20 "--config C3 --steps 20 --warmup 3" $REL ) > gpurun_out/r06_c3_check.txt 2>&1
cat gpurun_out/r06_c3_check.txt
