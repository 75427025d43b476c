for r in 1 2 3; do for cfg in "1 12" "4 12" "4 16" "2 16" "3 12"; do set -- $cfg
echo "group $1 workers $2: $(AB_REGISTER_GROUP=$1 AB_REGISTER_WORKERS=$2 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["stage_ms"]["register_63_frames_estimate_and_warp"], d["config"]["median"], d["config"]["registration"]["mean_inliers"])')"
done; done
