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

N > 1 (one process per GPU, torch.distributed over RCCL; without a launcher `--gpus N` spawns its own N ranks through
torch.distributed.run): the frame set is sharded by frame (BASELINE.json configs[3]): every rank holds its own 64 frames
of the same field, computes the per-pixel (sum, count) of its survivors, the partials are all-reduced over xGMI INSIDE
the library (ab_stack_sigma_clip_sharded: RCCL behind the C ABI) and divided, and the stretch runs on the result.  Work
per GPU is fixed, so scaling is "weak"; `value` counts the frame-pixels all ranks processed.
`--force-sharded` runs exactly that code path on one GPU (a one-rank communicator: real ncclAllReduce calls).
`--mode rowband` (N > 1) is the exact alternative: the same 64 frames on every rank, registration estimates sharded by
frame and exchanged, each rank warps / stacks / stretches its row band, statistics joined by histogram all-reduces;
total work is fixed, so scaling is "strong".

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
    ap.add_argument("--force-sharded", action="store_true",
                    help="1 GPU: run the N > 1 code path (partial -> RCCL all-reduce on a one-rank communicator -> finalize)")
    ap.add_argument("--mode", choices=("frames", "rowband"), default="frames",
                    help="N > 1: 'frames' = BASELINE configs[3] (frame shards, two-level, weak scaling); 'rowband' = exact row bands (strong scaling)")
    ap.add_argument("--config", choices=("C1", "C3", "C5"), default=None,
                    help="a BASELINE.json configuration other than the headline C2: one secondary JSON line (bench_configs.py)")
    ap.add_argument("--host-staged", action="store_true",
                    help="DRY RUN of the N > 1 launch line on ONE GPU: every rank on cuda:0, torch.distributed over gloo, the library's host-staged "
                         "communicator instead of RCCL (which refuses two ranks per device).  The line says so (config.transport); not a scaling figure")
    ap.add_argument("--host-planes", action="store_true",
                    help="also time the step with the 64 frames starting in pinned HOST memory (uploads pipelined with registration): config.host_resident_mpix_s")
    return ap.parse_args()


class stdout_to_stderr:
    """RCCL prints a version banner on fd 1 when a communicator is created; bench.py's stdout carries ONE JSON line"""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: re-run under torch.distributed.run with N ranks on this node."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def stack_source_hash():
    """sha256 of the stack kernel's sources: ties profiles/stack_pmc.json (rocprofv3 --pmc passes) to the code that was profiled"""
    import hashlib
    h = hashlib.sha256()
    for f in ("stack_sigma_clip.hip", "sortnet_gen.hpp", "sort_ops.hpp"):   # (sort_ops.hpp: where the network's instructions live)
        with open(os.path.join(ROOT, "astroburst_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


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
    if args.config:
        import bench_configs
        bench_configs.run(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)
    # stdout carries ONE JSON line: everything else that lands on fd 1 (RCCL prints a version banner there, at communicator
    # creation and again at exit) goes to stderr; the JSON is written to the saved descriptor at the very end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    wall = {"t0": time.perf_counter()}   # where the run's wall clock goes (the timed region is a fraction of a second of it)

    def wall_mark(name):
        now = time.perf_counter()
        wall[name] = round(now - wall.pop("t0"), 2)
        wall["t0"] = now
    import torch
    import torch.distributed as dist
    import astroburst_amd as ab
    from astroburst_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-GPU run as {args.gpus} GPUs")
    if torch.cuda.device_count() < 1 or (local_rank >= torch.cuda.device_count() and not args.host_staged):
        raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank} but {torch.cuda.device_count()} device(s) are visible")
    if args.host_staged:
        local_rank = 0   # every rank shares the one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            if args.host_staged:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    ctx = ab.Context(local_rank)
    ctx.use_torch_stream()
    name, cus, hbm = ctx.device_info()
    wall_mark("import_torch_and_library_s")
    sharded = world > 1 or args.force_sharded
    rowband = sharded and args.mode == "rowband"   # (with --force-sharded: one band = the whole image, the same code path)
    # the library's own communicator (RCCL behind the C ABI): rank 0 makes the id, torch.distributed carries it
    comm = None
    if sharded and args.host_staged:
        comm = ab.Comm.host(ctx, f"bench{os.environ.get('MASTER_PORT', '0')}", world, rank)
        comm.allreduce(torch.zeros(8, dtype=torch.float32, device=dev))
        torch.cuda.synchronize()
        assert comm.size == world and comm.rank == rank and comm.is_host
    elif sharded:
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(ab.Comm.unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(uid, 0)
        with stdout_to_stderr():
            comm = ab.Comm(ctx, bytes(uid.cpu().numpy().tobytes()), world, rank)
            comm.allreduce(torch.zeros(8, dtype=torch.float32, device=dev))   # first collective: connects the ranks
            torch.cuda.synchronize()
        assert comm.size == world and comm.rank == rank
    dev_names = [name]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, f"rank {rank}: cuda:{local_rank} {name}")
        dev_names = gathered

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
    row0, nrows = (0, R) if not rowband else ctx.shard_rows(R, world, rank)
    # Row bands INGEST bands ("GPU g owns rows [g R / G, (g + 1) R / G) of every frame (+ halo) ... each GPU ingests
    # N P / G pixels"): a rank keeps whole frames only for the reference and the targets it detects (k - 1 = rank mod G) and of
    # every target the rows its band of the output can read -- its rows + a halo sized from the generator's bounds (8 px shift, 0.5
    # degrees about the centre: |c| cols / 2 + 8 + 3 < 64 rows); the library checks every warp against what it really reads.
    kHalo = 64
    band0 = max(0, row0 - kHalo)
    band1 = min(R, row0 + nrows + kHalo)
    raw_band = [None] * N
    for k in range(N):
        # frame k sees the field through its own pointing error: a star at reference (x, y) lands at T_k(x, y)
        a_, b_, tx_, c_, d_, ty_ = transforms[k] if register else transforms[0]
        cat_k = (c_ * cx0 + d_ * cy0 + ty_, a_ * cx0 + b_ * cy0 + tx_, cflux)
        truth = torch.full((R, Cc), 200.0, dtype=torch.float32, device=dev) + synth.render_stars(R, Cc, cat_k, device=dev)
        border = 16 if k % 10 == 9 else 0
        frame = synth.make_frame(R, Cc, k + (0 if rowband else N * rank), device=dev, truth=truth, border=border)
        if rowband and world > 1 and k > 0:
            raw_band[k] = frame[band0:band1].clone()
            raw.append(frame if (k - 1) % world == rank else torch.empty((0, Cc), dtype=torch.float32, device=dev))
        else:
            raw.append(frame)
        del frame
    del truth
    resident_bytes = sum(t.numel() * 4 for t in raw) + sum(t.numel() * 4 for t in raw_band if t is not None)
    if rowband:   # this rank's rows of every registered frame; frame 0 (the reference) needs no warp: a view of its rows
        warped = [raw[0][row0:row0 + nrows]] + [torch.empty((nrows, Cc), dtype=torch.float32, device=dev) for _ in range(1, N)]
    else:
        warped = [raw[0]] + [torch.empty_like(raw[0]) for _ in range(1, N)] if register else raw
    stacked = torch.empty((nrows, Cc), dtype=torch.float32, device=dev)
    u8 = torch.empty((nrows, Cc), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    # the frames' ab_plane descriptors are marshalled ONCE, as a Rust / C host holds them: rebuilt per call they are 2 us of Python per
    # plane -- 0.4 ms per step with the GPU idle (round 6; the C ABI takes `const ab_plane *` either way)
    if not rowband and register and not args.known_transforms and not sharded:
        targets_pl, warped_out_pl, warped_pl = ctx.planes(raw[1:]), ctx.planes(warped[1:]), ctx.planes(warped)
    else:
        targets_pl, warped_out_pl, warped_pl = raw[1:], warped[1:], warped
    stack_ms, warp_ms, tail_ms, est_ms, kern_ms = [], [], [], [], []
    nsteps = args.steps + args.warmup
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(nsteps)]
    estimated = [None]

    def step(i):
        e = ev[i]
        e[4].record()
        if rowband:
            # estimates by frame across the ranks (frame k on rank (k - 1) mod G), exchanged as 80 bytes each; every rank then
            # warps ITS rows of every frame with the whole-image coordinates (bit-identical to the rows of a full warp)
            # Round 6: ONE call (ab_align_pairs_affine_rowband): a rank's own frames are warped -- its rows -- as they are fitted, overlapped
            # with the remaining estimates; then the exchange; then the other ranks' frames from this rank's rows + halo of them
            tg = [raw[k] if raw[k].numel() else raw_band[k] for k in range(1, N)]
            t0 = [0 if raw[k].numel() else band0 for k in range(1, N)]
            estimated[0] = ctx.align_pairs_affine_rowband(comm, raw[0], tg, warped[1:], row0, target_row0=t0, num_threads=8)
            e[0].record()
            e[1].record()
            ctx.stack_sigma_clip(warped, 3.0, 3.0, 5, out=stacked, want_rejected=False)
            e[2].record()
        else:
            if register and not args.known_transforms:
                # align_pair(frame 0, frame k, Affine) x 63 (pair.rs:41-77): estimate + warp, frame-parallel workers
                estimated[0] = ctx.align_pairs_affine(raw[0], targets_pl, warped_out_pl, num_threads=8)
            e[0].record()
            if register and args.known_transforms:
                for k in range(1, N):
                    ctx.warp_image(raw[k], transforms[k], R, Cc, out=warped[k])
            e[1].record()
            if not sharded:
                ctx.stack_sigma_clip(warped_pl, 3.0, 3.0, 5, out=stacked, want_rejected=False)
            else:  # per-GPU partial -> RCCL all-reduce(sum f64, count u32) over xGMI -> divide, all inside the library
                ctx.stack_sigma_clip_sharded(comm, warped, stacked, 3.0, 3.0, 5, want_rejected=False)
            e[2].record()
        # auto_stretch_preview (cmd/common.rs:18-22): statistics -> auto_stf -> apply_stf, one device chain, one sync
        _, st, _ = ctx.auto_stretch_preview(stacked, out=u8, comm=comm if rowband else None, total_rows=R)
        e[3].record()
        if i >= args.warmup:
            kern_ms.append(ctx.stack_last_kernel_ms())  # HIP events the library records right around its stack kernels
        return st

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall_mark("generate_frames_and_warmup_s")
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        st = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    wall_mark("timed_steps_s")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.host_staged else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    for i in range(args.warmup, args.warmup + args.steps):
        est_ms.append(ev[i][4].elapsed_time(ev[i][0]))
        warp_ms.append(ev[i][0].elapsed_time(ev[i][1]))
        stack_ms.append(ev[i][1].elapsed_time(ev[i][2]))
        tail_ms.append(ev[i][2].elapsed_time(ev[i][3]))
    rejected = ctx.last_rejected()

    ms_per_step = elapsed * 1e3 / args.steps
    job_frames = N if rowband else world * N               # rowband: the same N frames, split by rows (strong scaling)
    value = job_frames * P / 1e6 / (elapsed / args.steps)  # input MPix/s, whole job

    # ---- roofline of the north-star kernel (stack): algorithmic bytes = 4*N*P read + 4*P (12*P partial) written,
    # divided by the kernel's average duration between HIP events on the launch stream
    # stage time between this script's events (includes the host's launch overhead: the GPU is idle when the call starts)
    stage_stack_ms = sum(stack_ms) / len(stack_ms)
    # the kernels' own duration: events recorded by the library on the launch stream immediately around the launches
    stack_avg_ms = sum(kern_ms) / len(kern_ms)
    Pb = nrows * Cc                                  # pixels this rank stacks (a row band in rowband mode, else the image)
    out_bytes = 4 * Pb if not (sharded and not rowband) else 12 * Pb
    algo_bytes = 4 * N * Pb + out_bytes
    achieved = algo_bytes / (stack_avg_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes committed with the profile of this same command
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs; FETCH_SIZE doubled per the gfx950
    # calibration in profiles/: a loads-only kernel with this access pattern reads exactly 1/2)
    traffic = None
    profile_avg_ms = None
    try:
        with open(os.path.join(ROOT, "profiles", "stack_pmc.json")) as f:
            pmc = json.load(f)
        # only a counter taken from THIS kernel source counts: the profile script stores the hash of the stack kernel's sources
        if pmc.get("frames") == N and pmc.get("pixels") == P and pmc.get("kernel_source_sha256") == stack_source_hash() and not sharded:
            traffic = pmc["hbm_bytes_per_launch"]
            profile_avg_ms = pmc.get("profile_avg_ms")
    except Exception:
        pass
    # `frac` = `frac_in_step`: the kernels' duration between the library's HIP events inside the timed steps (the contract's
    # definition).  `frac_sustained`: the same launch repeated back to back (a VALU-bound kernel clocks lower when nothing idles
    # the chip between launches).  `frac_profile`: from the average under rocprofv3 in the committed profile of this source.
    # Round 4: `frac` / `achieved` are what the LAST COMMITTED rocprofv3 profile of this kernel source says (profiles/stack_pmc.json,
    # bound to the source by hash) when there is one, so that the line cannot lead with a better figure than the profile a reader
    # recomputes it from; the live in-step and back-to-back measurements of THIS run stand beside it.  Without a profile of this
    # source: the in-step measurement.
    achieved_profile = None if not profile_avg_ms else algo_bytes / (profile_avg_ms * 1e-3) / 1e9
    lead = achieved_profile if achieved_profile else achieved
    roofline = {"bound": "hbm", "kernel": "stack_sigma_clip_kernel<64> (fast pass + general pass over the deferred pixels = one stack launch)", "achieved": round(lead, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(lead / HBM_PEAK_GBS, 4),
                "frac_basis": "rocprofv3 average of the committed profile of this source (profiles/stack_pmc.json)" if achieved_profile else
                              "HIP events around the kernels inside this run's timed steps (no committed profile of this kernel source)",
                "achieved_in_step": round(achieved, 1),
                "frac_in_step": round(achieved / HBM_PEAK_GBS, 4), "frac_sustained": None,
                "frac_profile": None if not profile_avg_ms else round(algo_bytes / (profile_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "profile_avg_ms": profile_avg_ms,
                "algorithmic_bytes": algo_bytes, "avg_kernel_ms": round(stack_avg_ms, 4), "traffic": traffic,
                "back_to_back_ms": None}
    warp_avg_ms = (sum(warp_ms) / len(warp_ms)) if register else 0.0
    est_avg_ms = sum(est_ms) / len(est_ms)
    stage_ms = {"register_63_frames_estimate_and_warp" if not args.known_transforms else "register_estimate_skipped":
                round(est_avg_ms, 4), "warps_with_known_transforms": round(warp_avg_ms, 4),
                ("stack" if not sharded or rowband else "stack_partial_allreduce_finalize"): round(stage_stack_ms, 4),
                "stats_stf" + ("_hist_allreduce" if rowband else ""): round(sum(tail_ms) / len(tail_ms), 4)}
    stage_ms.pop("stack", None) if (sharded and not rowband) else None
    if sharded and not rowband:
        # inside the sharded stack of the LAST step (the library's own events): the partial stacks' span on its stream, and the time
        # inside the all-reduces + divisions, which run chunk by chunk on a second stream while the next chunk is stacked
        try:
            s_ms, c_ms = ctx.stack_sharded_last_ms()
            stage_ms["comm_overlapped_partial_stack_span"] = round(s_ms, 4)
            stage_ms["comm_allreduce_sum_count_and_divide"] = round(c_ms, 4)
        except Exception:
            pass
    if rowband:
        stage_ms = {"register_estimates_by_frame_own_rows_warped_as_fitted_exchange_other_frames_rows": round(est_avg_ms, 4),
                    "stack_own_rows": round(stage_stack_ms, 4), "stats_stf_hist_allreduce": round(sum(tail_ms) / len(tail_ms), 4)}
    # the warp is f64-VALU bound, not HBM bound: 112 f64 VALU instructions per pixel on its interior path (43 mul, 43 add, 4 fma,
    # 16 cvt f32->f64, 2 floor, 2 cvt ->i32, 1.5 cvt i32->f64, 1 cvt ->f32: the ISA of csrc/resample.hip's warp_kernel, round 4;
    # the kernel's time does not change when its loads are taken away, tools/time_warp.py with AB_ABLATE_WARP=2)
    warp_roofline = None
    if register and not rowband:
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
                         "f64_ops_per_pixel": 112, "achieved_Gops": round(112 * P / (per * 1e-3) / 1e9, 1),
                         "peak_Gops": 39300.0, "frac": round(112 * P / (per * 1e-3) / 1e9 / 39300.0, 4)}

    # the same stack launch back to back on the registered frames (no warps in between): isolates the kernel from the
    # clock / cache state the f64-heavy registration leaves behind
    iso_ms = None
    if world == 1 and not sharded:
        i0, i1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ctx.stack_sigma_clip(warped_pl, 3.0, 3.0, 5, out=stacked, want_rejected=False)
        i0.record()
        for _ in range(5):
            ctx.stack_sigma_clip(warped_pl, 3.0, 3.0, 5, out=stacked, want_rejected=False)
        i1.record()
        torch.cuda.synchronize()
        iso_ms = i0.elapsed_time(i1) / 5

    # ---- the same step with the frames starting in HOST memory, as a command of the application would hold them (Array2<f32> in
    # GLOBAL_IMAGE_CACHE): 64 x 67 MB cross PCIe once.  ONE registration call takes the 63 host frames (round 4): the library
    # uploads them on its own stream, an event behind each frame, and its pipeline registers a frame when it has landed, so the
    # link is busy from the first byte to the last and what remains after it is one group's registration.
    host_info = None
    if args.host_planes and world == 1 and not sharded and register and not args.known_transforms:
        hosts = [torch.empty((R, Cc), dtype=torch.float32, pin_memory=True) for _ in range(N)]
        for k in range(N):
            hosts[k].copy_(raw[k])
        up = [torch.empty_like(raw[0]) for _ in range(N)]
        hw = [up[0]] + [torch.empty_like(raw[0]) for _ in range(1, N)]
        copy_stream = torch.cuda.Stream()
        torch.cuda.synchronize()

        def upload_all():
            with torch.cuda.stream(copy_stream):
                for k in range(N):
                    up[k].copy_(hosts[k], non_blocking=True)

        def host_step():
            up[0].copy_(hosts[0], non_blocking=True)   # the reference frame: the stack reads it from HBM as well
            ctx.align_pairs_affine(up[0], hosts[1:], hw[1:], num_threads=8)
            ctx.stack_sigma_clip(hw, 3.0, 3.0, 5, out=stacked, want_rejected=False)
            ctx.auto_stretch_preview(stacked, out=u8)

        host_step()
        torch.cuda.synchronize()
        t_h = time.perf_counter()
        for _ in range(3):
            host_step()
        torch.cuda.synchronize()
        host_ms = (time.perf_counter() - t_h) / 3 * 1e3
        t_u = time.perf_counter()
        for _ in range(3):
            upload_all()
        torch.cuda.synchronize()
        up_ms = (time.perf_counter() - t_u) / 3 * 1e3
        same = all(bool(((hw[k] == warped[k]) | (hw[k].isnan() & warped[k].isnan())).all()) for k in (1, N // 2, N - 1))   # (NaN patches warp to NaN)
        host_info = {"host_resident_ms_per_step": round(host_ms, 3), "host_resident_mpix_s": round(N * P / 1e6 / (host_ms * 1e-3), 1),
                     "pcie_upload_only_ms": round(up_ms, 3), "pcie_GBs": round(4 * N * P / (up_ms * 1e-3) / 1e9, 1),
                     "over_pcie_bound": round(host_ms / up_ms, 3), "registered_frames_equal_device_resident_run": bool(same)}
        del hosts, up, hw

    roofline["back_to_back_ms"] = None if iso_ms is None else round(iso_ms, 4)
    roofline["frac_sustained"] = None if iso_ms is None else round(algo_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    # Round 6: the headline `frac` / `achieved` are measured LIVE IN THIS RUN -- the library's HIP events around
    # the stack launch (fast + general pass) inside the timed steps, the contract's definition -- not read from a committed file; the
    # same launch repeated back to back (`frac_sustained`: a VALU-saturated kernel clocks lower when nothing idles the chip between
    # launches) and the committed rocprofv3 profile of this kernel source (`frac_profile`) stand beside it.
    roofline["achieved"] = round(achieved, 1)
    roofline["frac"] = roofline["frac_in_step"]
    roofline["frac_basis"] = ("this run: HIP events on the library's stream around the stack launch (fast + general pass) inside the timed steps; "
                              "frac_sustained = the same launch back to back in this run, frac_profile = rocprofv3 average of the committed profile")
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

    wall_mark("roofline_extras_s")
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
        # the same oracle on ONE thread (SURVEY 8d asks for it beside the all-cores figure): a crop sized for ~3 s
        rows_1 = int(max(1, min(rows_s, 3.0 / max(per_row * threads, 1e-9))))
        crop1 = [c_[:rows_1] for c_ in crop]
        t1 = time.perf_counter()
        img1, _ = pyoracle.stack_images(crop1, 3.0, 3.0, 5, order=pyoracle.ORDER_SELECT, threads=1)
        cst1 = pyoracle.compute_image_stats(img1)
        pyoracle.apply_stf(img1, pyoracle.auto_stf(cst1), cst1, threads=1)
        dt1 = time.perf_counter() - t1
        cpu["one_thread"] = {"stack_stretch_only_mpix_s": round(N * rows_1 * Cc / 1e6 / dt1, 2), "cores": 1,
                             "sample": f"{N}x{rows_1}x{Cc} crop, {dt1:.1f} s"}
        # parity spot check of the timed configuration on that crop
        if sharded:   # the two-level estimator's own checker (one shard = all of this rank's frames)
            ps, pc, _ = pyoracle.stack_partial(crop, 3.0, 3.0, 5)
            img = np.where(pc > 0, (ps / np.maximum(pc, 1)).astype(np.float32), np.float32(0))
        got = stacked[:rows_s].cpu().numpy()
        bad = int((~((got == img) | (np.isnan(got) & np.isnan(img)))).sum())
        rel = float(np.nanmax(np.abs(got - img) / np.maximum(np.abs(img), 1e-30)))
        cpu["parity_vs_gpu"] = {"pixels": int(img.size), "bit_mismatches": bad, "max_rel_err": rel,
                                "checker": "oracle.stack_partial + divide (two-level)" if sharded else "oracle.stack_images (reference's single-level)"}

    if rank == 0:
        # the whole step against HBM: every byte the step's stages must move at least once, over the step's wall time
        # stack + statistics and stretch (one kernel that reads the plane once and writes the u8 plane: 5 P; the row-band chain makes
        # five passes: 16 P + 5 P) + unfused warps
        step_bytes = (4 * N * P + 4 * P) + (21 if rowband else 5) * P + (8 * P * (N - 1) if register else 0)
        step_gbs = step_bytes / (ms_per_step * 1e-3) / 1e9 * (1 if rowband else world)
        out = {
            "metric": "MPix/s sigma-clipped stack+stretch, 64x4096x4096 f32",
            "value": round(value, 1), "unit": "MPix/s (input frame-pixels)", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if rowband else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"C2: {N}x{R}x{Cc} f32 synthetic frames per GPU: " if not rowband else
                                    f"C2 split by rows over {world} GPUs: {N}x{R}x{Cc} f32 synthetic frames in all: ")
                                   + ((f"affine register ({N - 1} x align_channel_affine estimate + bicubic warp) + "
                                       if not args.known_transforms else
                                       f"bicubic warp with the {N - 1} generating transforms (estimation skipped: diagnostic) + ")
                                      if register else "")
                                   + ("kappa-sigma stack (3/3/5)" if not sharded else
                                      "exact kappa-sigma stack of each GPU's row band" if rowband else
                                      "per-GPU kappa-sigma partial + RCCL all-reduce(sum f64, count u32) inside the library + divide"
                                      + (" [--force-sharded: one-rank communicator]" if world == 1 else ""))
                                   + " + image stats + auto-STF u8 (auto_stretch_preview)"
                                   + (" with histogram all-reduces" if rowband else ""),
                       "frames_per_gpu": N, "rows": R, "cols": Cc, "device": name, "cus": cus,
                       "rccl_ranks": (comm.size if comm and not comm.is_host else 0), "comm_ranks": (comm.size if comm else 0),
                       "rccl_collectives_per_step": (comm.collectives_issued // nsteps if comm and not comm.is_host else 0),
                       "comm_collectives_per_step": (comm.collectives_issued // nsteps if comm else 0),
                       "devices": dev_names, "mode": ("rowband" if rowband else "frames") if sharded else "single",
                       "resident_input_bytes_rank0": resident_bytes,
                       **({"rowband_ingest": f"rank 0 holds the reference, {sum(1 for t in raw[1:] if t.numel())} whole targets (the ones it detects) and rows "
                                             f"[{band0}, {band1}) of all {N - 1} targets (its {nrows} rows + a {kHalo}-row halo): "
                                             f"{resident_bytes / 1e9:.3f} GB against {N * P * 4 / 1e9:.3f} GB for the frame set"} if rowband and world > 1 else {}),
                       **({"transport": f"HOST-STAGED DRY RUN: {world} ranks share ONE GPU (cuda:0), collectives through shared memory, not RCCL / xGMI: "
                                        "the launch line and the sharded code path executed end to end, not a scaling measurement"} if args.host_staged else {}),
                       "output_mpix_per_s": round((1 if rowband else world) * P / 1e6 / (elapsed / args.steps), 1),
                       "rejected_pixels": rejected, "median": st.median,
                       "measured_copy_GBs": round(copy_gbs, 1), "stage_ms": stage_ms, "registration": reg_info,
                       "fallbacks": ctx.fallback_counts(), "library": ab.version() if hasattr(ab, "version") else None,
                       **({"host_resident_mpix_s": host_info["host_resident_mpix_s"], "host_planes": host_info} if host_info else {})},
            "roofline": roofline,
            "roofline_step": {"bound": "hbm", "achieved": round(step_gbs, 1), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                              "frac": round(step_gbs / (HBM_PEAK_GBS * world), 4), "algorithmic_bytes": step_bytes,
                              "note": "whole step (register + stack + stats + STF) bytes / ms_per_step: the registration estimate is latency / VALU bound, not HBM bound"},
            "roofline_warp": warp_roofline,
            "cpu_baseline": cpu,
        }
        wall_mark("cpu_baseline_s")
        wall.pop("t0", None)
        out["wall_s"] = wall   # (a fresh box pages the image in during the first `import torch`: one to two minutes on its own)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
