"""Developer tool: decode rate and fused raw stacking vs decode + stack at 64 x 4096^2 (run on the GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import astroburst_amd as ab  # noqa: E402

R = C = 4096
N = 64
ctx = ab.Context(0)
ctx.use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(5)


def timed(fn, label, reps=5, bytes_moved=None):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    extra = f"  {bytes_moved / ms / 1e6:8.1f} GB/s" if bytes_moved else ""
    print(f"{label:64s} {ms:8.3f} ms{extra}")


i16, f32 = [], []
for _ in range(N):
    v = (torch.randn((R, C), device="cuda", generator=g) * 40.0).round().clamp(-32768, 32767)
    vi = v.to(torch.int16).to(torch.int32)
    i16.append((((vi & 0xFF) << 8) | ((vi >> 8) & 0xFF)).to(torch.int16).view(torch.uint8).reshape(-1))
    fb = (v + 32768.0).to(torch.float32).view(torch.int32)
    f32.append((((fb & 0xFF) << 24) | (((fb >> 8) & 0xFF) << 16) | (((fb >> 16) & 0xFF) << 8) | ((fb >> 24) & 0xFF)).view(torch.uint8).reshape(-1))
out = torch.empty((R, C), device="cuda")
dec = [torch.empty((R, C), device="cuda") for _ in range(N)]
P = R * C
timed(lambda: ctx.fits_decode_pixels(i16[0], R, C, 16, 1.0, 32768.0, out=dec[0]), "decode one BITPIX 16 plane (BZERO 32768)", bytes_moved=6 * P)
timed(lambda: ctx.fits_decode_pixels(f32[0], R, C, -32, out=dec[0]), "decode one BITPIX -32 plane", bytes_moved=8 * P)
timed(lambda: [ctx.fits_decode_pixels(i16[k], R, C, 16, 1.0, 32768.0, out=dec[k]) for k in range(N)] and
      ctx.stack_sigma_clip(dec, 3.0, 3.0, 5, out=out, want_rejected=False), "decode 64 BITPIX 16 planes, then stack")
timed(lambda: ctx.stack_sigma_clip_raw(i16, R, C, 16, 1.0, 32768.0, out=out, want_rejected=False), "fused: stack 64 raw BITPIX 16 planes")
timed(lambda: ctx.stack_sigma_clip_raw(f32, R, C, -32, out=out, want_rejected=False), "fused: stack 64 raw BITPIX -32 planes")
timed(lambda: ctx.stack_sigma_clip(dec, 3.0, 3.0, 5, out=out, want_rejected=False), "stack 64 decoded f32 planes")
