cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests/test_gpu_detect_affine.py tests/test_gpu_subframe.py tests/test_gpu_full_size.py -m gpu -x -q < /dev/null > gpurun_out/r04s_pytest.log 2>&1
tail -n 4 gpurun_out/r04s_pytest.log
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['config']['stage_ms'], d['config']['registration'])"; done
