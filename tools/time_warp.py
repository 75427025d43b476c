"""Developer tool: time warp_image alone on a 4096^2 device plane (HIP events on torch's stream), for a few transforms."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from astroburst_amd import Context  # noqa: E402

R = C = 4096
ctx = Context(0)
ctx.use_torch_stream()
g = torch.Generator(device="cuda").manual_seed(3)
src = torch.rand((R, C), device="cuda", generator=g) * 1000.0
out = torch.empty_like(src)
ts = bench.rigid_transforms(8, R, C)
cases = {"bench frame 1": ts[1], "bench frame 5": ts[5], "shift only": (1.0, 0.0, 2.25, 0.0, 1.0, -1.5),
         "rotation 0.05 rad": (0.99875, -0.04998, 110.0, 0.04998, 0.99875, -95.0)}
for name, t in cases.items():
    for _ in range(3):
        ctx.warp_image(src, t, R, C, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ctx.warp_image(src, t, R, C, out=out)
    e1.record()
    torch.cuda.synchronize()
    print(f"warp_image 4096^2 {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  {dict((k, v) for k, v in os.environ.items() if k.startswith('AB_'))}")
