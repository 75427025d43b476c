"""Developer tool: latency of ONE registration call over a few device-resident 4096^2 frames (the tail of a host-fed batch)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from astroburst_amd import Context, synth  # noqa: E402

ctx = Context(0)
R = C = 4096
y, x, flux = synth.star_catalog(R, C, 6000, seed=1)
cat = (y, x, flux * 30.0)
frames = [synth.make_frame(R, C, k, cat=cat, shift=(0.7 * k, -0.4 * k), device="cuda") for k in range(9)]
outs = [torch.empty_like(f) for f in frames]
for n in (1, 2, 3, 4, 8):
    ctx.align_pairs_affine(frames[0], frames[1:1 + n], outs[1:1 + n], num_threads=8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.align_pairs_affine(frames[0], frames[1:1 + n], outs[1:1 + n], num_threads=8)
    torch.cuda.synchronize()
    print(f"{n} target(s): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call")
    if n >= 4:
        os.environ["AB_TRACE"] = "1"
