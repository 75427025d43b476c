#!/bin/bash
# round 6: the bicubic kernels' pieces in COLUMN-major runs per XCD (default) against row-major ids (variant): C2 and C3, interleaved
mkdir -p gpurun_out
DEV=$PWD/astroburst_amd/libastroburst_hip_dev.so
VAR=$PWD/astroburst_amd/csrc/build/variants/libab_rowmajor.so
timeout 900 python -m pytest tests -m gpu -x -q -k "warp or resample or affine or rowband or shift or align" > gpurun_out/r06w_tests.log 2>&1; echo "warp tests rc=$?" > gpurun_out/r06w_rc.txt
tail -2 gpurun_out/r06w_tests.log
run() { tag=$1; cfg="$2"; lib=$3; AB_LIB_PATH=$lib timeout 600 python bench.py $cfg --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); s=d["config"]["stage_ms"]; print(sys.argv[1], sys.argv[2], d["ms_per_step"], [v for k,v in s.items() if k.startswith("register")])' "$tag" "$cfg"; }
( echo "# bench.py, developer builds, interleaved: column-major runs per XCD (default) vs row-major piece ids (-DAB_WARP_ROW_MAJOR)"
  for i in 1 2 3 4; do run runs "" $DEV; run rowmajor "" $VAR; done
  for i in 1 2 3; do run runs "--config C3" $DEV; run rowmajor "--config C3" $VAR; done ) > gpurun_out/r06w_ab2.txt 2>&1
cat gpurun_out/r06w_ab2.txt
