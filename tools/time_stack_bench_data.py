"""Developer tool: stack kernel time on the bench's own raw / registered frames vs max_iterations (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import astroburst_amd as ab  # noqa: E402
from astroburst_amd import synth  # noqa: E402
from bench import rigid_transforms  # noqa: E402

R = C = 4096
N = 64
P = R * C
ctx = ab.Context(0)
ctx.use_torch_stream()
cy, cx, cf = synth.star_catalog(R, C, int(360.0 * P / 1e6))
cf = cf * 25.0
T = rigid_transforms(N, R, C)
raw = []
for k in range(N):
    a_, b_, tx_, c_, d_, ty_ = T[k]
    truth = torch.full((R, C), 200.0, device="cuda") + synth.render_stars(R, C, (c_ * cx + d_ * cy + ty_, a_ * cx + b_ * cy + tx_, cf), device="cuda")
    raw.append(synth.make_frame(R, C, k, device="cuda", truth=truth, border=16 if k % 10 == 9 else 0))
warped = [raw[0]] + [torch.empty_like(raw[0]) for _ in range(1, N)]
ctx.align_pairs_affine(raw[0], raw[1:], warped[1:], 8)
out = torch.empty((R, C), device="cuda")
if len(sys.argv) > 2 and sys.argv[1] == "--dump":   # raw f32 planes for tools/stack_ablate.hip --dir
    os.makedirs(sys.argv[2], exist_ok=True)
    for k, w in enumerate(warped):
        w.cpu().numpy().tofile(os.path.join(sys.argv[2], f"frame_{k:02d}.f32"))
    print("dumped", N, "frames to", sys.argv[2])
    sys.exit(0)


def t(frames, label, it=5):
    out = torch.empty_like(frames[0])
    ctx.stack_sigma_clip(frames, 3.0, 3.0, it, out=out, want_rejected=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ctx.stack_sigma_clip(frames, 3.0, 3.0, it, out=out, want_rejected=False)
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:46s} it={it}  {e0.elapsed_time(e1) / 5:7.3f} ms   kernels {ctx.stack_last_kernel_ms():6.3f} ms   rejected/px {ctx.last_rejected() / frames[0].numel():.3f}")


for it in (1, 2, 3, 5):
    t(warped, "registered frames", it)
t(raw, "raw (unregistered) frames")
inner = [w[64:-64, 64:-64].contiguous() for w in warped]
t(inner, "registered, 64 px border cropped away")
clean = [torch.nan_to_num(w, nan=1200.0) for w in warped]
t(clean, "registered, NaN patches filled")
nn = int(sum(int(torch.isnan(w).sum()) for w in warped))
zz = int(sum(int((w == 0).sum()) for w in warped))
print("NaN samples", nn, " zero samples", zz, " of", N * P)
