import sys, numpy as np
sys.path.insert(0, ".")
from astroburst_amd import Context
from astroburst_amd.core import BatchStackConfig
ctx = Context(0)
rng = np.random.default_rng(0)
shape = (120, 173)
for n in (33, 40, 64):
    lights = [rng.normal(400 + 30 * k, 12, shape).astype(np.float32) for k in range(n)]
    flat = rng.normal(1.0, 0.05, shape).astype(np.float32)
    for norm in (False, True):
        for f in (None, flat):
            print("n", n, "norm", norm, "flat", f is not None, flush=True)
            out = ctx.run_batch_channel(lights, None, None, f, BatchStackConfig(normalize_before_stack=norm))
            print("  ok", out[2], flush=True)
