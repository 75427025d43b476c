"""Developer tool (GPU box): does a frame's registration depend on the batch it is registered in?  align_pairs_affine over all 63 targets
against the same targets in chunks of 16 (what bench.py --host-planes does)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from astroburst_amd import Context, synth  # noqa: E402

N, R, C = 64, 4096, 4096
ctx = Context(0)
ctx.use_torch_stream()
cy0, cx0, cf = synth.star_catalog(R, C, max(8, int(360.0 * R * C / 1e6)))
cflux = cf * 25.0
T = bench.rigid_transforms(N, R, C)
raw = []
for k in range(N):
    a_, b_, tx_, c_, d_, ty_ = T[k]
    truth = torch.full((R, C), 200.0, device="cuda") + synth.render_stars(R, C, (c_ * cx0 + d_ * cy0 + ty_, a_ * cx0 + b_ * cy0 + tx_, cflux), device="cuda")
    raw.append(synth.make_frame(R, C, k, device="cuda", truth=truth, border=16 if k % 10 == 9 else 0))
w_all = [torch.empty_like(raw[0]) for _ in range(N - 1)]
res_all = ctx.align_pairs_affine(raw[0], raw[1:], w_all, num_threads=8)
torch.cuda.synchronize()
w_ch = [torch.empty_like(raw[0]) for _ in range(N - 1)]
res_ch = []
for c0 in range(0, N, 16):
    tg = [k for k in range(c0, min(c0 + 16, N)) if k != 0]
    res_ch += ctx.align_pairs_affine(raw[0], [raw[k] for k in tg], [w_ch[k - 1] for k in tg], num_threads=8)
torch.cuda.synchronize()
bad = 0
for k in range(N - 1):
    same_t = res_all[k].transform == res_ch[k].transform
    same_w = bool(((w_all[k] == w_ch[k]) | (w_all[k].isnan() & w_ch[k].isnan())).all())   # (NaN patches warp to NaN)
    if not (same_t and same_w):
        bad += 1
        d = max(abs(a - b) for a, b in zip(res_all[k].transform, res_ch[k].transform))
        print(f"frame {k + 1}: transform equal {same_t} (max coeff diff {d:.3e}), inliers {res_all[k].inliers} / {res_ch[k].inliers}, method {res_all[k].method} / {res_ch[k].method}, "
              f"warped equal {same_w} ({int(((w_all[k] != w_ch[k]) & ~(w_all[k].isnan() & w_ch[k].isnan())).sum())} px differ)")
print("frames that differ between the whole batch and chunks of 16:", bad)
