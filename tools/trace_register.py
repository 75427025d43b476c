"""Developer tool: AB_TRACE=1 stage stamps of a few serial align_channel_affine calls (run on the GPU box)."""
import os
import sys

os.environ["AB_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import astroburst_amd as ab  # noqa: E402
from astroburst_amd import synth  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = ab.Context(0)
ctx.use_torch_stream()
y, x, flux = synth.star_catalog(R, R, 1500, seed=3)
cat = (y, x, flux * 25.0)
frames = [synth.make_frame(R, R, k, cat=cat, device="cuda", shift=(0.37 * k % 7 - 3, 3 - 0.53 * k % 6), bad_patch_rate=0.0) for k in range(6)]
torch.cuda.synchronize()
for k in range(1, 6):
    sys.stderr.write(f"--- pair {k}\n")
    ctx.align_channel_affine(frames[0], frames[k], 8)
