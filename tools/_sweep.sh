for c in 4 8 16 32 64; do for w in 12 16; do
echo "chunk $c workers $w: $(AB_TILE_CHUNK=$c AB_REGISTER_WORKERS=$w timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["stage_ms"]["register_63_frames_estimate_and_warp"])')"
done; done
