"""Developer tool: timings of the SURVEY 8(f) rows 3 and 4 on 4096^2 device planes (subframe scoring, previews, tiles, IPC buffer)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from astroburst_amd import Context, synth  # noqa: E402
from astroburst_amd.core import ImageStats, StfParams  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


ctx = Context(0)
R = C = 4096
dev = torch.device("cuda", 0)
cy0, cx0, cf = synth.star_catalog(R, C, max(8, int(360.0 * R * C / 1e6)))
frames = []
for k in range(16):
    truth = torch.full((R, C), 200.0, dtype=torch.float32, device=dev) + synth.render_stars(R, C, (cy0, cx0, cf * 25.0), device=dev)
    frames.append(synth.make_frame(R, C, k, device=dev, truth=truth))
torch.cuda.synchronize()
t = timed(lambda: ctx.analyze_subframes(frames), reps=3)
print(f"analyze_subframes, 16 x 4096^2: {t:.2f} ms ({t / 16:.2f} ms per frame)")
g = torch.Generator(device="cuda").manual_seed(9)
r, gg, b = (torch.rand((R, C), device="cuda", generator=g) for _ in range(3))
st = [ImageStats(min=0.0, max=1.0, median=0.3, mad=0.1, sigma=0.15, mean=0.4, valid_count=R * C)] * 3
stf = [StfParams(0.05, 0.3, 1.0)] * 3
print(f"render_rgb_preview 4096^2 -> 2048^2 with STF: {timed(lambda: ctx.render_rgb_preview(r, gg, b, 2048, stf, st)):.3f} ms")
print(f"render_rgb_preview 4096^2 -> 4096^2 with STF: {timed(lambda: ctx.render_rgb_preview(r, gg, b, 4096, stf, st)):.3f} ms")
print(f"generate_tile_pyramid_rgb (5 levels, 341 tiles, STF): {timed(lambda: ctx.generate_tile_pyramid_rgb(r, gg, b, 256, stf, st)):.3f} ms")
print(f"generate_tile_pyramid mono (percentile bounds + 5 levels): {timed(lambda: ctx.generate_tile_pyramid(gg, 256)):.3f} ms")
print(f"ipc_encode_with_header full 4096^2: {timed(lambda: ctx.ipc_encode_with_header(gg, 0)):.3f} ms; downsampled to 2048: {timed(lambda: ctx.ipc_encode_with_header(gg, 2048)):.3f} ms")
