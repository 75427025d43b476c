"""Developer tool: what makes the stack kernel slower on the bench's frames than on the ablation tool's? (run on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import astroburst_amd as ab  # noqa: E402
from astroburst_amd import synth  # noqa: E402

R = C = 4096
N = 64
ctx = ab.Context(0)
ctx.use_torch_stream()
out = torch.empty((R, C), device="cuda")


def t(frames, label):
    ctx.stack_sigma_clip(frames, 3.0, 3.0, 5, out=out, want_rejected=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ctx.stack_sigma_clip(frames, 3.0, 3.0, 5, out=out, want_rejected=False)
    e1.record()
    torch.cuda.synchronize()
    print(f"{label:58s} {e0.elapsed_time(e1) / 5:7.3f} ms   rejected/px {ctx.last_rejected() / (R * C):.3f}")


g = torch.Generator(device="cuda").manual_seed(1)
gauss = [1200.0 + 12.7 * torch.randn((R, C), device="cuda", generator=g) for _ in range(N)]
t(gauss, "torch-allocated pure Gaussian frames")
big = torch.empty((N, R, C), device="cuda")
for k in range(N):
    big[k] = gauss[k]
t([big[k] for k in range(N)], "same, one contiguous 64 x 64 MiB allocation")
pad = torch.empty((N, R * C + 4096), device="cuda")
fr = []
for k in range(N):
    v = pad[k, 1024 * (k % 4):1024 * (k % 4) + R * C].view(R, C)
    v.copy_(gauss[k])
    fr.append(v)
t(fr, "same, frames 64 MiB + 16 KiB apart (staggered)")
cr = [torch.where(torch.rand((R, C), device="cuda", generator=g) < 1e-4, f * 30.0, f) for f in gauss]
t(cr, "+ cosmic rays 1e-4")
truth = torch.full((R, C), 200.0, device="cuda") + synth.render_stars(R, C, tuple(x * (25.0 if i == 2 else 1.0) for i, x in enumerate(synth.star_catalog(R, C, 6039))), device="cuda")
raw = [synth.make_frame(R, C, k, device="cuda", truth=truth, border=16 if k % 10 == 9 else 0) for k in range(N)]
t(raw, "bench-like frames (stars, shot noise, CR, NaN patches, borders)")
nb = [synth.make_frame(R, C, k, device="cuda", truth=truth, border=0, bad_patch_rate=0.0) for k in range(N)]
t(nb, "bench-like without NaN patches / borders")
flat = torch.full((R, C), 200.0, device="cuda")
nf = [synth.make_frame(R, C, k, device="cuda", truth=flat, border=0, bad_patch_rate=0.0) for k in range(N)]
t(nf, "bench-like without stars, patches, borders")

# registered frames carry interpolation-dependent noise levels: emulate with per-frame sigma in [0.6, 1.0] x 12.7
het = [1200.0 + 12.7 * (0.6 + 0.4 * ((k * 7) % 11) / 10.0) * torch.randn((R, C), device="cuda", generator=g) for k in range(N)]
for it in (1, 2, 3, 4, 5):
    ctx.stack_sigma_clip(het, 3.0, 3.0, it, out=out, want_rejected=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ctx.stack_sigma_clip(het, 3.0, 3.0, it, out=out, want_rejected=False)
    e1.record()
    torch.cuda.synchronize()
    print(f"heteroscedastic frames, max_iterations={it}: {e0.elapsed_time(e1) / 5:7.3f} ms   rejected/px {ctx.last_rejected() / (R * C):.3f}")
