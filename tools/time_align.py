"""Developer tool: time the registration estimators on 4096^2 device frames (run on the GPU box)."""
import sys
import time

sys.path.insert(0, ".")
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
