"""Time the batch calibration channel on 64 x 4096^2 device-resident lights: the fused channel, its two
kernels, and the unfused three-step route.  HIP events on the context's stream (torch's current stream)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from astroburst_amd import Context  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def main():
    import os
    n, rows, cols = int(os.environ.get("N_FRAMES", "64")), 4096, 4096
    ctx = Context(0)
    g = torch.Generator(device="cuda").manual_seed(11)
    bias = torch.randn((rows, cols), device="cuda", generator=g) * 2.0 + 100.0
    flat = torch.randn((rows, cols), device="cuda", generator=g) * 0.05 + 1.0
    lights = []
    for k in range(n):
        f = torch.randn((rows, cols), device="cuda", generator=g) * 12.0 + (400.0 + 5.0 * k)
        f[torch.rand((rows, cols), device="cuda", generator=g) < 1e-3] += 900.0
        lights.append(f * flat + bias)
    torch.cuda.synchronize()
    gb = n * rows * cols * 4 / 1e9
    t = timed(lambda: ctx.run_batch_channel(lights, bias=bias, flat=flat))
    print(f"fused channel (means + stack + stats): {t:.2f} ms   ({2 * gb / t:.2f} TB/s over two reads of the lights)")
    norm = ctx.normalize_frames([ctx.calibrate_light(l, bias=bias, flat=flat) for l in lights])
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        ctx.sigma_clipped_mean_stack(norm)
    ev1.record()
    torch.cuda.synchronize()
    print(f"sigma_clipped_mean_stack, GPU time between events: {ev0.elapsed_time(ev1) / 5:.2f} ms per call")
    t2 = timed(lambda: ctx.sigma_clipped_mean_stack(norm))
    print(f"sigma_clipped_mean_stack alone: {t2:.2f} ms   ({gb / t2:.2f} TB/s, {gb / t2 / 8.0:.2f} of the HBM roofline)")
    from astroburst_amd.core import BatchStackConfig
    import os
    if os.environ.get("AB_BATCH_STAGE"):
        print("stage", os.environ["AB_BATCH_STAGE"], "max_iterations=0:", f"{timed(lambda: ctx.sigma_clipped_mean_stack(norm, BatchStackConfig(max_iterations=0))):.2f} ms")
        return
    for it in (0, 1, 2, 3):
        ti = timed(lambda: ctx.sigma_clipped_mean_stack(norm, BatchStackConfig(max_iterations=it)))
        print(f"  max_iterations={it}: {ti:.2f} ms")
    t3 = timed(lambda: ctx.run_batch_channel(lights, bias=bias, flat=flat, config=BatchStackConfig(normalize_before_stack=False)))
    print(f"fused channel without normalisation (one read): {t3:.2f} ms")

    def unfused():
        cal = [ctx.calibrate_light(l, bias=bias, flat=flat) for l in lights]
        nf = ctx.normalize_frames(cal)
        return ctx.sigma_clipped_mean_stack(nf)
    del norm
    t4 = timed(unfused, reps=2)
    print(f"unfused calibrate -> normalize -> stack: {t4:.2f} ms")


if __name__ == "__main__":
    main()
