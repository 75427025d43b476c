cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py --host-planes --no-cpu-baseline > gpurun_out/r04l_bench_host.json 2> gpurun_out/r04l_bench_host.err < /dev/null
python -c "
import json;d=json.loads(open('gpurun_out/r04l_bench_host.json').read().strip().splitlines()[-1]);print(d['config']['host_planes'], d['config']['stage_ms'])"
for q in 2 4 8 16 24; do
echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['config']['stage_ms'])"
done
for w in 8 16 20; do
echo "GPU_MAX_HW_QUEUES=16 AB_REGISTER_WORKERS=$w"; AB_REGISTER_WORKERS=$w GPU_MAX_HW_QUEUES=16 timeout 600 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['ms_per_step'], d['config']['stage_ms'])"
done
