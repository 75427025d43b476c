"""Developer tool: median_combine (calibration masters) of deep 4096^2 stacks (run on the GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import astroburst_amd as ab
ctx = ab.Context(0)
for n in [int(v) for v in os.environ.get("N_LIST", "90,128,200,256").split(",")]:
    fr = [1000 + 10 * torch.randn((4096, 4096), device="cuda") for _ in range(n)]
    ctx.median_combine(fr); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): ctx.median_combine(fr)
    torch.cuda.synchronize()
    print(f"median_combine {n} x 4096^2: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
    del fr
