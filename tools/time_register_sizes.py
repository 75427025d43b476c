"""Developer tool: ab_align_pairs_affine (63 targets) at several frame sizes with the same star count -- is the registration
stage bound by the GPU (time ~ pixels) or by per-frame host / runtime work (time ~ constant)?  Also one frame alone, serially."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import astroburst_amd as ab  # noqa: E402
from astroburst_amd import synth  # noqa: E402

N = int(os.environ.get("N_FRAMES", "64"))
ctx = ab.Context(0)
ctx.use_torch_stream()
for R in [int(a) for a in (sys.argv[1:] or ["1024", "2048", "4096"])]:
    y, x, flux = synth.star_catalog(R, R, 1500, seed=3)
    cat = (y, x, flux * 25.0)
    frames = [synth.make_frame(R, R, k, cat=cat, device="cuda", shift=(0.37 * k % 7 - 3, 3 - 0.53 * k % 6), bad_patch_rate=0.0) for k in range(N)]
    out = [torch.empty_like(f) for f in frames[1:]]
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        res = ctx.align_pairs_affine(frames[0], frames[1:], out, num_threads=8)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    for k in range(1, 9):
        ctx.align_channel_affine(frames[0], frames[k], 8)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / 8 * 1e3
    print(f"{R}x{R}: align_pairs_affine({N - 1} targets) {dt:8.2f} ms = {dt / (N - 1) * 1e3:7.1f} us per frame; one align_channel_affine alone {one:6.2f} ms", flush=True)
