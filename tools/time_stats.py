"""Times auto_stretch_preview (statistics -> auto_stf -> u8 stretch, one synchronisation) on a device plane with both engines
of the histogram path: the resident kernel (csrc/stats_resident.hpp) and the five-pass chain (AB_STATS_CHAIN=1)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import astroburst_amd as ab  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    cols = int(sys.argv[2]) if len(sys.argv) > 2 else rows
    rng = np.random.default_rng(1)
    img = (1000.0 + 30.0 * rng.standard_normal((rows, cols))).astype(np.float32)
    stars = rng.random((rows, cols)) < 1e-3
    img[stars] += rng.pareto(2.5, stars.sum()).astype(np.float32) * 2000.0
    d = torch.from_numpy(img).cuda()
    u8 = torch.empty((rows, cols), dtype=torch.uint8, device="cuda")
    ctx = ab.Context(0)
    res = {}
    for engine in ("resident", "chain", "resident", "chain"):
        os.environ["AB_STATS_CHAIN"] = "1" if engine == "chain" else "0"
        for _ in range(5):
            _, st, p = ctx.auto_stretch_preview(d, out=u8)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            _, st, p = ctx.auto_stretch_preview(d, out=u8)
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        t0 = time.perf_counter()
        for _ in range(30):
            ctx.compute_image_stats(d)
        only = (time.perf_counter() - t0) / 30 * 1e3
        key = (st.median, st.mad, st.mean, int(u8.to(torch.int64).sum().item()))
        res.setdefault(engine, key)
        print(f"{engine:9s} {rows}x{cols}: preview median {ts[len(ts) // 2]:.4f} ms (min {ts[0]:.4f}); compute_image_stats alone {only:.4f} ms; "
              f"median {st.median!r} mad {st.mad!r}")
    same = res["resident"][:2] == res["chain"][:2] and res["resident"][3] == res["chain"][3]
    print("engines agree (median, mad, u8 checksum):", same, "| mean differs by", abs(res["resident"][2] - res["chain"][2]))


if __name__ == "__main__":
    main()
