#!/usr/bin/env python3
"""bench.py -- the north-star metric on MI355X.

  python bench.py --gpus N --steps K --warmup W

One STEP = one pass of the hot path over one device-resident synthetic stack of
64 x 4096 x 4096 f32 frames (BASELINE.json configs[1]):
   star-based affine registration of 63 frames on frame 0: align_channel_affine per frame (normalise, detect,
   triangle votes, RANSAC) -> bicubic warp_image with the ESTIMATED transform
   -> per-pixel kappa-sigma stack (3 sigma / 3 sigma / 5 iterations)
   -> compute_image_stats -> auto_stf -> apply_stf (u8)
all through the C ABI of libastroburst_hip.so, inputs already in HBM when the clock starts.

N > 1 (one process per GPU, torch.distributed over RCCL): the frame set is sharded by frame
(BASELINE.json configs[3]): every rank holds its own 64 frames of the same field, computes the
per-pixel (sum, count) of its survivors, the partials are all-reduced over xGMI and divided, and
the stretch runs on the result.  Work per GPU is fixed, so scaling is "weak"; `value` counts the
frame-pixels all ranks processed.

Prints ONE JSON line (rank 0).  `value` = input MPix/s = frames x pixels / second for the whole
step; `roofline` prices the stacking kernel alone against HBM; `cpu_baseline` times the CPU
oracle (a C restatement of the reference's Rust path, "port") on a bounded crop of the same frames.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--cols", type=int, default=4096)
    ap.add_argument("--no-register", action="store_true", help="skip registration (stack + stretch only)")
    ap.add_argument("--known-transforms", action="store_true",
                    help="warp with the generating transforms instead of estimating them (diagnostic, not the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration")
    ap.add_argument("--cpu-reg-frames", type=int, default=4, help="frame pairs the CPU baseline registers (each ~2 s on 8 threads)")
    return ap.parse_args()


def rigid_transforms(n, rows, cols, seed=7):
    """per-frame rigid pointing errors: rotation U(-0.05, 0.05) deg about the frame centre, shift U(-8, 8) px.
    T maps reference (output) coordinates to frame (source) coordinates, as AffineTransform::map does."""
    import random
    rnd = random.Random(seed)
    ts = [(1.0, 0.0, 0.0, 0.0, 1.0, 0.0)]
    cx, cy = (cols - 1) / 2.0, (rows - 1) / 2.0
    for _ in range(1, n):
        ang = math.radians(rnd.uniform(-0.05, 0.05))
        c, s_ = math.cos(ang), math.sin(ang)
        dx, dy = rnd.uniform(-8, 8), rnd.uniform(-8, 8)
        ts.append((c, -s_, cx - c * cx + s_ * cy + dx, s_, c, cy - s_ * cx - c * cy + dy))
    return ts


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import astroburst_amd as ab
    from astroburst_amd import synth
    from astroburst_amd.distributed import sharded_stack

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = ab.Context(local_rank)
    ctx.use_torch_stream()
    name, cus, hbm = ctx.device_info()

    N, R, Cc = args.frames, args.rows, args.cols
    P = R * Cc
    # ---- synthetic, device-resident input (each rank: its own N frames, seeds offset by rank) ----
    # a rich field (360 stars / Mpix, Pareto fluxes x 25): the reference's normalize_for_detection clips at the
    # 99.9th percentile, which wipes out star contrast unless > 0.1 % of the pixels belong to stars -- on sparser
    # fields both the reference and this library fall through to phase correlation (see tests)
    n_stars = max(8, int(360.0 * P / 1e6))
    cy_, cx_, cf_ = synth.star_catalog(R, Cc, n_stars)
    cat = (cy_, cx_, cf_ * 25.0)
    transforms = rigid_transforms(N, R, Cc)
    register = not args.no_register
    cy0, cx0, cflux = cat
    raw = []
    for k in range(N):
        # frame k sees the field through its own pointing error: a star at reference (x, y) lands at T_k(x, y)
        a_, b_, tx_, c_, d_, ty_ = transforms[k] if register else transforms[0]
        cat_k = (c_ * cx0 + d_ * cy0 + ty_, a_ * cx0 + b_ * cy0 + tx_, cflux)
        truth = torch.full((R, Cc), 200.0, dtype=torch.float32, device=dev) + synth.render_stars(R, Cc, cat_k, device=dev)
        border = 16 if k % 10 == 9 else 0
        raw.append(synth.make_frame(R, Cc, k + N * rank, device=dev, truth=truth, border=border))
    del truth
    warped = [raw[0]] + [torch.empty_like(raw[0]) for _ in range(1, N)] if register else raw
    stacked = torch.empty((R, Cc), dtype=torch.float32, device=dev)
    u8 = torch.empty((R, Cc), dtype=torch.uint8, device=dev)
    if world > 1:
        psum = torch.empty((R, Cc), dtype=torch.float64, device=dev)
        pcnt = torch.empty((R, Cc), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    stack_ms, warp_ms, tail_ms, est_ms, kern_ms = [], [], [], [], []
    nsteps = args.steps + args.warmup
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(nsteps)]
    estimated = [None]

    def step(i):
        e = ev[i]
        e[4].record()
        if register and not args.known_transforms:
            # align_pair(frame 0, frame k, Affine) x 63 (pair.rs:41-77): estimate + warp, frame-parallel workers
            estimated[0] = ctx.align_pairs_affine(raw[0], raw[1:], warped[1:], num_threads=8)
        e[0].record()
        if register and args.known_transforms:
            for k in range(1, N):
                ctx.warp_image(raw[k], transforms[k], R, Cc, out=warped[k])
        e[1].record()
        if world == 1:
            ctx.stack_sigma_clip(warped, 3.0, 3.0, 5, out=stacked, want_rejected=False)
            e[2].record()
        else:
            def partial_fn(fr):
                r = ctx.stack_partial_into(fr, psum, pcnt)
                e[2].record()
                return r

            sharded_stack(warped, partial_fn, lambda s_, c_: ctx.stack_finalize_partial_into(s_, c_, stacked))
        st = ctx.compute_image_stats(stacked)          # syncs the stream: the stack's events below are complete
        if i >= args.warmup:
            kern_ms.append(ctx.stack_last_kernel_ms())  # HIP events the library records right around its stack kernels
        p = ctx.auto_stf(st)
        ctx.apply_stf(stacked, p, st, out=u8)
        e[3].record()
        return st

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        st = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for i in range(args.warmup, args.warmup + args.steps):
        est_ms.append(ev[i][4].elapsed_time(ev[i][0]))
        warp_ms.append(ev[i][0].elapsed_time(ev[i][1]))
        stack_ms.append(ev[i][1].elapsed_time(ev[i][2]))
        tail_ms.append(ev[i][2].elapsed_time(ev[i][3]))
    rejected = ctx.last_rejected()

    ms_per_step = elapsed * 1e3 / args.steps
    value = world * N * P / 1e6 / (elapsed / args.steps)  # input MPix/s, whole job

    # ---- roofline of the north-star kernel (stack): algorithmic bytes = 4*N*P read + 4*P (12*P partial) written,
    # divided by the kernel's average duration between HIP events on the launch stream
    # stage time between this script's events (includes the host's launch overhead: the GPU is idle when the call starts)
    stage_stack_ms = sum(stack_ms) / len(stack_ms)
    # the kernels' own duration: events recorded by the library on the launch stream immediately around the launches
    stack_avg_ms = sum(kern_ms) / len(kern_ms)
    out_bytes = 4 * P if world == 1 else 12 * P
    algo_bytes = 4 * N * P + out_bytes
    achieved = algo_bytes / (stack_avg_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes committed with the profile of this same command
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs; FETCH_SIZE doubled per the gfx950
    # calibration in profiles/: a loads-only kernel with this access pattern reads exactly 1/2)
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "stack_pmc.json")) as f:
            pmc = json.load(f)
        if pmc.get("frames") == N and pmc.get("pixels") == P:
            traffic = pmc["hbm_bytes_per_launch"]
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": "stack_sigma_clip_kernel<64> (fast pass + general pass over the deferred pixels = one stack launch)", "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "algorithmic_bytes": algo_bytes, "avg_kernel_ms": round(stack_avg_ms, 4), "traffic": traffic,
                "back_to_back_ms": None}
    warp_avg_ms = (sum(warp_ms) / len(warp_ms)) if register else 0.0
    est_avg_ms = sum(est_ms) / len(est_ms)
    stage_ms = {"register_63_frames_estimate_and_warp" if not args.known_transforms else "register_estimate_skipped":
                round(est_avg_ms, 4), "warps_with_known_transforms": round(warp_avg_ms, 4),
                "stack": round(stage_stack_ms, 4),
                "stats_stf" + ("_allreduce" if world > 1 else ""): round(sum(tail_ms) / len(tail_ms), 4)}
    # the warp is f64-VALU bound (the reference's f64 bicubic, ~111 f64 ops per pixel), not HBM bound
    warp_roofline = None
    if register:
        if args.known_transforms:
            per = warp_avg_ms / (N - 1)
        else:  # the warps ran inside align_pairs_affine, overlapped with detection: time the kernel on its own here
            w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            scratch = torch.empty_like(raw[0])
            ctx.warp_image(raw[1], estimated[0][0].transform, R, Cc, out=scratch)
            w0.record()
            for k in range(1, 9):
                ctx.warp_image(raw[k], estimated[0][k - 1].transform, R, Cc, out=scratch)
            w1.record()
            torch.cuda.synchronize()
            per = w0.elapsed_time(w1) / 8
            del scratch
        warp_roofline = {"bound": "valu_f64", "kernel": "warp_kernel", "avg_kernel_ms": round(per, 4),
                         "hbm_GBs": round(8 * P / (per * 1e-3) / 1e9, 1),
                         "f64_ops_per_pixel": 111, "achieved_Gops": round(111 * P / (per * 1e-3) / 1e9, 1),
                         "peak_Gops": 39300.0, "frac": round(111 * P / (per * 1e-3) / 1e9 / 39300.0, 4)}

    # the same stack launch back to back on the registered frames (no warps in between): isolates the kernel from the
    # clock / cache state the f64-heavy registration leaves behind
    iso_ms = None
    if world == 1:
        i0, i1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ctx.stack_sigma_clip(warped, 3.0, 3.0, 5, out=stacked, want_rejected=False)
        i0.record()
        for _ in range(5):
            ctx.stack_sigma_clip(warped, 3.0, 3.0, 5, out=stacked, want_rejected=False)
        i1.record()
        torch.cuda.synchronize()
        iso_ms = i0.elapsed_time(i1) / 5

    roofline["back_to_back_ms"] = None if iso_ms is None else round(iso_ms, 4)
    # measured streaming ceiling of this GPU (float4 copy), for context
    a = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    ctx.bench_copy(a, b)
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(10):
        ctx.bench_copy(a, b)
    c1.record()
    torch.cuda.synchronize()
    copy_gbs = 10 * 2 * a.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9
    del a, b

    # how well the estimated transforms reproduce the generating ones (registration sanity, not a parity claim)
    reg_info = None
    if register and estimated[0] is not None:
        errs = []
        for k, r in enumerate(estimated[0], start=1):
            t, g = r.transform, transforms[k]
            # displacement error at the frame corners and centre
            for (x, y) in ((0.0, 0.0), (Cc - 1.0, 0.0), (0.0, R - 1.0), (Cc - 1.0, R - 1.0), ((Cc - 1) / 2.0, (R - 1) / 2.0)):
                ex = (t[0] - g[0]) * x + (t[1] - g[1]) * y + (t[2] - g[2])
                ey = (t[3] - g[3]) * x + (t[4] - g[4]) * y + (t[5] - g[5])
                errs.append(math.hypot(ex, ey))
        methods = {}
        for r in estimated[0]:
            methods[r.method] = methods.get(r.method, 0) + 1
        reg_info = {"methods": methods, "max_err_px_vs_generating_transform": round(max(errs), 4),
                    "mean_inliers": round(sum(r.inliers for r in estimated[0]) / len(estimated[0]), 1)}

    # ---- CPU baseline: the oracle (C restatement of the reference, OpenMP) on a bounded crop ----
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import pyoracle
        import numpy as np
        threads = pyoracle.max_threads()
        # registration leg: the oracle's align_channel_affine + warp_image on ONE whole frame pair, scaled to N - 1
        t_reg = 0.0
        reg_parity = None
        if register and not args.known_transforms:
            ref_h = raw[0].cpu().numpy()
            n_reg = max(1, min(N - 1, args.cpu_reg_frames))
            for k in range(1, n_reg + 1):
                tgt_h = raw[k].cpu().numpy()
                t1 = time.perf_counter()
                want = pyoracle.align_channel_affine(ref_h, tgt_h, num_threads=8)
                pyoracle.warp_image(tgt_h, want.transform, R, Cc)
                t_reg += time.perf_counter() - t1
                if k == 1:
                    got = estimated[0][0]
                    reg_parity = {"method": [got.method, want.method], "inliers": [got.inliers, want.inliers],
                                  "max_abs_coeff_diff": float(max(abs(a - b) for a, b in zip(got.transform, want.transform)))}
            t_reg_total, t_reg = t_reg, t_reg / n_reg
            del ref_h, tgt_h
        probe_rows = 4 * threads if 4 * threads < R else R     # a few rows per thread: a fair rate estimate
        crop = [w[:probe_rows].cpu().numpy() for w in warped]
        t1 = time.perf_counter()
        pyoracle.stack_images(crop, 3.0, 3.0, 5, order=pyoracle.ORDER_SELECT)
        per_row = (time.perf_counter() - t1) / probe_rows
        rows_s = int(max(probe_rows, min(R, args.cpu_seconds / max(per_row, 1e-9))))
        crop = [w[:rows_s].cpu().numpy() for w in warped]
        t1 = time.perf_counter()
        img, _ = pyoracle.stack_images(crop, 3.0, 3.0, 5, order=pyoracle.ORDER_SELECT)
        cst = pyoracle.compute_image_stats(img)
        pyoracle.apply_stf(img, pyoracle.auto_stf(cst), cst)
        dt = time.perf_counter() - t1
        t_full = dt * R / rows_s + t_reg * (N - 1)      # whole-step estimate from the two bounded samples
        cpu = {"value": round(N * P / 1e6 / t_full, 2), "unit": "MPix/s", "cores": threads, "kind": "port",
               "sample": f"oracle/liboracle.so (C restatement, OpenMP): kappa-sigma stack + stats + auto-STF on a "
                         f"{N}x{rows_s}x{Cc} crop of the registered frames ({dt:.1f} s, scaled to {R} rows)"
                         + (f" + align_channel_affine + warp_image of {n_reg} {R}x{Cc} frame pairs ({t_reg_total:.1f} s, "
                            f"scaled to {N - 1} frames; detection's labelling is serial as in the reference)"
                            if t_reg else ""),
               "stack_stretch_only_mpix_s": round(N * rows_s * Cc / 1e6 / dt, 2)}
        if reg_parity:
            cpu["registration_parity_frame1"] = reg_parity
        # parity spot check of the timed configuration on that crop
        got = stacked[:rows_s].cpu().numpy()
        bad = int((~((got == img) | (np.isnan(got) & np.isnan(img)))).sum())
        rel = float(np.nanmax(np.abs(got - img) / np.maximum(np.abs(img), 1e-30)))
        cpu["parity_vs_gpu"] = {"pixels": int(img.size), "bit_mismatches": bad, "max_rel_err": rel}

    if rank == 0:
        out = {
            "metric": "MPix/s sigma-clipped stack+stretch, 64x4096x4096 f32",
            "value": round(value, 1), "unit": "MPix/s (input frame-pixels)", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C2: {N}x{R}x{Cc} f32 synthetic frames per GPU: "
                                   + ((f"affine register ({N - 1} x align_channel_affine estimate + bicubic warp) + "
                                       if not args.known_transforms else
                                       f"bicubic warp with the {N - 1} generating transforms (estimation skipped: diagnostic) + ")
                                      if register else "")
                                   + ("kappa-sigma stack (3/3/5)" if world == 1 else
                                      "per-GPU kappa-sigma partial + RCCL all-reduce(sum f64, count i32) + divide")
                                   + " + image stats + auto-STF u8",
                       "frames_per_gpu": N, "rows": R, "cols": Cc, "device": name, "cus": cus,
                       "output_mpix_per_s": round(world * P / 1e6 / (elapsed / args.steps), 1),
                       "rejected_pixels": rejected, "median": st.median,
                       "measured_copy_GBs": round(copy_gbs, 1), "stage_ms": stage_ms, "registration": reg_info},
            "roofline": roofline,
            "roofline_warp": warp_roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
