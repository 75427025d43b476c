"""Developer tool: time ab_align_pairs_affine on the bench's 64 x 4096^2 frames (one reference, 63 targets).  A/B friendly:
run it several times in one gpurun call with different env knobs (AB_REGISTER_WORKERS)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from astroburst_amd import Context, synth  # noqa: E402

N, R, C = 64, 4096, 4096
dev = torch.device("cuda", 0)
ctx = Context(0)
ctx.use_torch_stream()
cy0, cx0, cf = synth.star_catalog(R, C, max(8, int(360.0 * R * C / 1e6)))
cflux = cf * 25.0
transforms = bench.rigid_transforms(N, R, C)
raw = []
for k in range(N):
    a_, b_, tx_, c_, d_, ty_ = transforms[k]
    cat_k = (c_ * cx0 + d_ * cy0 + ty_, a_ * cx0 + b_ * cy0 + tx_, cflux)
    truth = torch.full((R, C), 200.0, dtype=torch.float32, device=dev) + synth.render_stars(R, C, cat_k, device=dev)
    raw.append(synth.make_frame(R, C, k, device=dev, truth=truth, border=16 if k % 10 == 9 else 0))
warped = [torch.empty_like(raw[0]) for _ in range(1, N)]
torch.cuda.synchronize()
NOALIGN = os.environ.get("NOALIGN") == "1"     # estimates only (register_frames): no warps
def call():
    if NOALIGN:
        return ctx.register_frames(raw[0], raw[1:], num_threads=8)
    return ctx.align_pairs_affine(raw[0], raw[1:], warped, num_threads=8)
for _ in range(2):
    res = call()
torch.cuda.synchronize()
ts = []
for _ in range(int(os.environ.get('REPS', '8'))):
    t0 = time.perf_counter()
    res = call()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
knobs = {k: v for k, v in os.environ.items() if k.startswith("AB_")}
print(f"{'register_frames' if NOALIGN else 'align_pairs_affine'} x63: min {min(ts):.2f} ms, median {sorted(ts)[len(ts) // 2]:.2f} ms  {knobs}  methods {sorted(set(r.method for r in res))}  all " + " ".join(f"{t:.2f}" for t in ts))
