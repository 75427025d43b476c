"""Developer tool: time the registration estimators on 4096^2 device frames (run on the GPU box)."""
import sys
import time

import os  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import astroburst_amd as ab  # noqa: E402
from astroburst_amd import synth  # noqa: E402

R = C = 4096
ctx = ab.Context(0)
ctx.use_torch_stream()
y, x, flux = synth.star_catalog(R, C, 6000, seed=3)
cat = (y, x, flux * 25.0)
ref = synth.make_frame(R, C, 0, cat=cat, device="cuda", bad_patch_rate=0.0)
tgt = synth.make_frame(R, C, 1, cat=cat, device="cuda", shift=(3.3, -5.1), bad_patch_rate=0.0)
torch.cuda.synchronize()
for name, fn in [("phase_correlate", lambda: ctx.phase_correlate(ref, tgt)),
                 ("detect_stars(5 sigma)", lambda: ctx.detect_stars(ref, 5.0)[0][:1]),
                 ("align_channel_affine", lambda: ctx.align_channel_affine(ref, tgt, 8))]:
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / 3 * 1e3:9.2f} ms   ->", out)
stars, m, s = ctx.detect_stars(ref, 5.0)
print("stars:", len(stars), "bg", m, s)

# finer breakdown of align_channel_affine (affine.rs:129-212): normalise -> detect -> match
import numpy as np  # noqa: E402


def timeit(name, fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print(f"  {name:34s} {(time.perf_counter() - t0) / reps * 1e3:9.2f} ms")
    return out


print("breakdown:")
norm_r = timeit("normalize_for_detection(ref)", lambda: ctx.normalize_for_detection(ref))
norm_t = ctx.normalize_for_detection(tgt)
sr = timeit("detect_stars(norm ref, 3.5 sigma)", lambda: ctx.detect_stars(norm_r, 3.5))[0]
st = ctx.detect_stars(norm_t, 3.5)[0]
print("  stars on normalised frames:", len(sr), len(st))
rxy = np.array([(s.x, s.y) for s in sr])
txy = np.array([(s.x, s.y) for s in st])
timeit("affine_from_stars (host)", lambda: ctx.affine_from_stars(rxy, txy, R, C, 8))
timeit("estimate_background(ref, 256)", lambda: ctx.estimate_background(ref, 256))
timeit("compute_image_stats(ref)", lambda: ctx.compute_image_stats(ref))
out = torch.empty_like(ref)
timeit("warp_image", lambda: ctx.warp_image(tgt, (1.0, 0.0, 5.1, 0.0, 1.0, -3.3), R, C, out=out))

# batch registration of a 64-frame stack (the bench's register stage)
tg = [synth.make_frame(R, C, k, cat=cat, device="cuda", shift=(0.37 * k - 3.0, 2.0 - 0.21 * k), bad_patch_rate=0.0) for k in range(1, 16)]
torch.cuda.synchronize()
res = timeit("register_frames(ref, 15 targets)", lambda: ctx.register_frames(ref, tg, 8), reps=2)
print("  methods:", sorted({r.method for r in res}), " first:", [round(v, 3) for v in res[0].transform])
