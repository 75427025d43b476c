"""Developer tool: time the > 64-frame stack (csrc/stack_wide.hip, one wave per pixel) on 4096^2 device frames."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from astroburst_amd import Context  # noqa: E402

ctx = Context(0)
SIDE = int(os.environ.get("SIDE", "4096"))
g = torch.Generator(device="cuda").manual_seed(1)
for n in [int(v) for v in os.environ.get("N_LIST", "64,96,128,256").split(",")]:
    fr = [torch.randn((SIDE, SIDE), device="cuda", generator=g) * 15.0 + 1200.0 for _ in range(n)]
    out = torch.empty((SIDE, SIDE), device="cuda")
    median = os.environ.get("MODE", "") == "median"  # median_combine (calibration masters) instead of the kappa-sigma stack
    run = (lambda: ctx.median_combine(fr, out=out)) if median else (lambda: ctx.stack_sigma_clip(fr, out=out, want_rejected=False))
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"{n:4d} frames x {SIDE}^2{' median' if median else ''}: {ms:8.2f} ms  ({n * SIDE * SIDE * 4 / ms / 1e9:.2f} TB/s of samples)  kernels {ctx.stack_last_kernel_ms():.2f} ms")
    del fr
