#!/bin/bash
# round 6: four adjacent output pixels per lane in the warp (default) against round 5's two pixels 256 apart (developer build, AB_WARP_TWO=1)
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
timeout 1200 python -m pytest tests -m gpu -x -q -k "warp or resample or affine or rowband or shift or align or golden or full_size" > gpurun_out/r06k_tests.log 2>&1; echo "warp tests rc=$?" > gpurun_out/r06k_rc.txt
tail -2 gpurun_out/r06k_tests.log
run() { tag=$1; cfg="$2"; shift 2; env AB_LIB_PATH=$DEV "$@" timeout 600 python bench.py $cfg --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); s=d["config"]["stage_ms"]; print(sys.argv[1], sys.argv[2], d["ms_per_step"], [v for k,v in s.items() if k.startswith("register")])' "$tag" "$cfg"; }
( echo "# bench.py, developer build, interleaved: four adjacent pixels per lane (default) vs two pixels 256 apart (AB_WARP_TWO=1)"
  for i in 1 2 3 4; do run four "" AB_X=0; run two "" AB_WARP_TWO=1; done
  for i in 1 2; do run four "--config C3" AB_X=0; run two "--config C3" AB_WARP_TWO=1; done
  echo "## tools/time_warp.py"
  AB_LIB_PATH=$DEV timeout 300 python tools/time_warp.py 2>&1 | grep -v amdgpu | head -8
  AB_LIB_PATH=$DEV AB_WARP_TWO=1 timeout 300 python tools/time_warp.py 2>&1 | grep -v amdgpu | head -8 ) > gpurun_out/r06_warp4_ab.txt 2>&1
cat gpurun_out/r06_warp4_ab.txt
