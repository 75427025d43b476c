for r in 1 2; do for w in 8 12 16 20; do
echo "workers $w: $(AB_REGISTER_WORKERS=$w timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["stage_ms"]["register_63_frames_estimate_and_warp"])')"
done
echo "hwq8 w12: $(GPU_MAX_HW_QUEUES=8 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["stage_ms"]["register_63_frames_estimate_and_warp"])')"
echo "hwq8 w16: $(GPU_MAX_HW_QUEUES=8 AB_REGISTER_WORKERS=16 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["stage_ms"]["register_63_frames_estimate_and_warp"])')"
echo "nowarpstream: $(AB_NO_WARP_STREAM=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["stage_ms"]["register_63_frames_estimate_and_warp"])')"
done
