# same-box A/B of library variants on the whole bench step: AB_LIB_PATH selects the library; rounds interleaved
for r in 1 2 3; do for v in "$@"; do
echo "$v: $(AB_LIB_PATH=$GRAFT_REPO_ROOT/build/$v timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["stage_ms"]["register_63_frames_estimate_and_warp"], d["config"]["median"])')"
done; done
