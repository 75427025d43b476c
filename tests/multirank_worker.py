"""One rank of a multi-process run of csrc/sharded.hip + csrc/comm.hip on ONE GPU (spawned by tests/test_gpu_multirank.py).

    python tests/multirank_worker.py <comm name> <nranks> <rank> <scenario> <out dir>

Every rank creates its own ab_ctx on cuda:0 and joins the HOST-STAGED communicator /abcomm_<name> (RCCL refuses two ranks on
one device); from there on the calls are exactly what a multi-GPU host makes: ab_register_frames_sharded,
ab_stack_sigma_clip_rowband + ab_allgather_rows, ab_compute_image_stats_sharded, ab_auto_stretch_preview(comm),
ab_stack_sigma_clip_sharded.  Inputs are generated from seeds, identically on every rank and in the parent test; results go
to <out dir>/rank<r>.npz for the parent to hold against the oracle.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def star_frames():
    """(reference, 5 targets) of the registration scenario -- the parent regenerates the same frames"""
    import torch
    from astroburst_amd import synth
    rows, cols = 512, 640
    y, x, flux = synth.star_catalog(rows, cols, 400, seed=13)
    cat = (y, x, flux * 30.0)
    ref = synth.make_frame(rows, cols, 0, cat=cat, bad_patch_rate=0.0, cosmic_rate=0.0)
    shifts = [(2.5, -1.0), (-3.0, 4.0), (0.5, 0.25), (6.0, -5.5)]
    tgts = [synth.make_frame(rows, cols, k + 1, cat=cat, shift=s, bad_patch_rate=0.0, cosmic_rate=0.0) for k, s in enumerate(shifts)]
    tgts.append(torch.full((rows, cols), 1000.0))  # no stars: phase correlation / identity
    return ref, tgts


def stack_frames(n=16, rows=50, cols=96):
    from astroburst_amd import synth
    return synth.make_stack(n, rows, cols)


def big_image(rows=2100, cols=2000, seed=5):
    """> 4 000 000 pixels: the histogram path of compute_image_stats (stats.rs:18-22)"""
    rng = np.random.default_rng(seed)
    img = rng.normal(1200.0, 30.0, size=(rows, cols)).astype(np.float32)
    img[rng.random((rows, cols)) < 1e-3] *= 20.0
    img[5:9, 100:400] = np.nan
    img[rows - 3:, :50] = 0.0
    return img


def res_tuple(r):
    return np.array(list(r.transform) + [float(r.matched_stars), float(r.inliers), r.residual_px, float(hash(r.method) % 1000)], dtype=np.float64)


def plane(t):
    """ab_plane of a (possibly empty) CUDA tensor"""
    import ctypes as C
    from astroburst_amd._lib import Plane
    return Plane(C.c_void_p(t.data_ptr() if t.numel() else 0), t.shape[0], t.shape[1], 1)


def scenario_all(ab, ctx, comm, rank, nranks, out):
    import torch
    res = {}
    # ---- raw collectives: sums across ranks, larger than one staging window (4 MiB), odd sizes --------------------------
    n = 3_000_017
    t = (torch.arange(n, dtype=torch.float64, device="cuda") * (rank + 1))
    comm.allreduce(t, "sum")
    tri = nranks * (nranks + 1) // 2
    assert torch.equal(t, torch.arange(n, dtype=torch.float64, device="cuda") * tri)
    u = torch.full((70001,), rank + 5, dtype=torch.int32, device="cuda")
    comm.allreduce(u, "max")
    assert int(u.min()) == nranks + 4 and int(u.max()) == nranks + 4
    w = torch.full((9,), 10 - rank, dtype=torch.int64, device="cuda")
    comm.allreduce(w, "min")
    assert int(w.max()) == 10 - (nranks - 1)
    # f64 sums are formed in rank order on every rank: bit-identical everywhere
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    f = torch.randn(100_003, dtype=torch.float64, generator=g).cuda()
    comm.allreduce(f, "sum")
    res["f64_sum"] = f.cpu().numpy()

    # ---- ab_register_frames_sharded: target i on rank i mod size, all results on every rank ---------------------------
    ref, tgts = star_frames()
    regs = ctx.register_frames_sharded(comm, ref.cuda(), [x.cuda() for x in tgts], num_threads=8)
    res["reg_transform"] = np.array([r.transform for r in regs])
    res["reg_counts"] = np.array([[r.matched_stars, r.inliers] for r in regs], dtype=np.int64)
    res["reg_residual"] = np.array([r.residual_px for r in regs])
    res["reg_method"] = np.array([r.method for r in regs])

    # ---- row bands (exact): this rank's band of the stack of ALL frames, the rejected count summed, bands gathered --------
    for tag, (rows, cols) in {"band": (50, 96), "thin": (2, 70)}.items():  # "thin": more ranks than rows can fill -> empty bands
        host = stack_frames(16 if tag == "band" else 8, rows, cols)
        dev = [x.cuda() for x in host]
        row0, nrows = ctx.shard_rows(rows, nranks, rank)
        band = torch.empty((nrows, cols), device="cuda")
        _, rej = ctx.stack_sigma_clip_rowband(comm, dev, band)
        full = torch.zeros((rows, cols), device="cuda")
        ctx.allgather_rows(comm, band, full)
        res[f"{tag}_full"] = full.cpu().numpy()
        res[f"{tag}_rej"] = np.array([rej], dtype=np.int64)
        res[f"{tag}_rows"] = np.array([row0, nrows], dtype=np.int64)

    # ---- ab_compute_image_stats_sharded / ab_auto_stretch_preview(comm): histogram path and exact path ----------------
    import ctypes as C
    from astroburst_amd._lib import ImageStatsC
    for tag, img in {"hist": big_image(), "exact": big_image(300, 500, seed=6), "tiny": big_image(2, 700, seed=8)}.items():
        rows, cols = img.shape
        row0, nrows = ctx.shard_rows(rows, nranks, rank)
        band = torch.from_numpy(np.ascontiguousarray(img[row0:row0 + nrows])).cuda()
        pb = plane(band)
        s = ImageStatsC()
        ctx.use_torch_stream()
        ctx._check(ctx._L.ab_compute_image_stats_sharded(ctx._h, comm._h, C.byref(pb), rows, C.byref(s)))
        res[f"stats_{tag}"] = np.array([s.min, s.max, s.median, s.mad, s.sigma, s.mean, float(s.valid_count)])
        if nrows > 0:
            u8, st, stf = ctx.auto_stretch_preview(band, comm=comm, total_rows=rows)
            res[f"u8_{tag}"] = u8.cpu().numpy()
            res[f"stf_{tag}"] = np.array([stf.shadow, stf.midtone, stf.highlight])
        else:  # an empty band still takes part in the statistics' collectives
            from astroburst_amd._lib import AutoStfConfigC, StfParamsC
            cfg = AutoStfConfigC(0.25, -2.8)
            p = StfParamsC()
            ctx._check(ctx._L.ab_auto_stretch_preview(ctx._h, comm._h, C.byref(pb), rows, C.byref(cfg), None, C.byref(s), C.byref(p)))
            res[f"u8_{tag}"] = np.zeros((0, cols), dtype=np.uint8)
            res[f"stf_{tag}"] = np.array([p.shadow, p.midtone, p.highlight])

    # ---- frame shards (BASELINE configs[3]): partial over THIS rank's frames -> all-reduce(sum, count) -> divide ----------
    host = stack_frames(24, 97, 160)
    f0, nf = ctx.shard_frames(24, nranks, rank)
    dev = [x.cuda() for x in host[f0:f0 + nf]]
    outp = torch.empty((97, 160), device="cuda")
    _, rej = ctx.stack_sigma_clip_sharded(comm, dev, outp, want_rejected=True)
    res["frames_out"] = outp.cpu().numpy()
    res["frames_rej"] = np.array([rej], dtype=np.int64)
    res["collectives"] = np.array([comm.collectives_issued], dtype=np.int64)
    np.savez(os.path.join(out, f"rank{rank}.npz"), **res)


def scenario_fail(ab, ctx, comm, rank, nranks, out):
    """one rank fails locally -> EVERY rank returns an error, nobody hangs, and the communicator stays usable"""
    import torch
    from astroburst_amd._lib import AB_ERR_CANCELLED, AB_ERR_COMM, AB_ERR_INVALID
    codes = []
    ref, tgts = star_frames()
    bad = nranks - 1
    # (1) a cancel on one rank during registration
    if rank == bad:
        ctx.request_cancel()
    try:
        ctx.register_frames_sharded(comm, ref.cuda(), [x.cuda() for x in tgts[:4]], num_threads=8)
        codes.append(0)
    except ab.AstroBurstError as e:
        codes.append(e.code)
    ctx.clear_cancel()
    # (2) an empty frame shard on one rank of the two-level stack
    host = stack_frames(6, 20, 64)
    dev = [x.cuda() for x in host] if rank != bad else []
    outp = torch.empty((20, 64), device="cuda")
    try:
        if dev:
            ctx.stack_sigma_clip_sharded(comm, dev, outp)
        else:  # the wrapper refuses an empty list: call the entry point itself with n_local = 0
            import ctypes as C
            from astroburst_amd._lib import Plane, StackConfig
            cfg = StackConfig(3.0, 3.0, 5, 0)
            po = plane(outp)
            ctx.use_torch_stream()
            ctx._check(ctx._L.ab_stack_sigma_clip_sharded(ctx._h, comm._h, (Plane * 1)(), 0, C.byref(cfg), C.byref(po), None))
        codes.append(0)
    except ab.AstroBurstError as e:
        codes.append(e.code)
    # (3) a band of the wrong size on one rank of the row-band stack
    host = stack_frames(8, 24, 64)
    dev = [x.cuda() for x in host]
    row0, nrows = ctx.shard_rows(24, nranks, rank)
    band = torch.empty((nrows + (1 if rank == bad else 0), 64), device="cuda")
    try:
        ctx.stack_sigma_clip_rowband(comm, dev, band)
        codes.append(0)
    except ab.AstroBurstError as e:
        codes.append(e.code)
    # (4) the communicator is still in step: a plain all-reduce and a correct row-band stack work
    t = torch.full((1000,), float(rank + 1), dtype=torch.float64, device="cuda")
    comm.allreduce(t, "sum")
    ok = bool(torch.all(t == nranks * (nranks + 1) / 2))
    band = torch.empty((nrows, 64), device="cuda")
    _, rej = ctx.stack_sigma_clip_rowband(comm, dev, band)
    want = [AB_ERR_CANCELLED, AB_ERR_INVALID if rank == bad else AB_ERR_COMM, AB_ERR_INVALID if rank == bad else AB_ERR_COMM]
    np.savez(os.path.join(out, f"rank{rank}.npz"), codes=np.array(codes), want=np.array(want), ok=np.array([ok]), rej=np.array([rej]))


def scenario_die(ab, ctx, comm, rank, nranks, out):
    """the last rank dies without a word; the others must come back with AB_ERR_COMM within their timeout"""
    import torch
    comm.set_timeout_ms(3000)
    t = torch.ones(1000, dtype=torch.float64, device="cuda")
    comm.allreduce(t, "sum")  # everybody is here
    if rank == nranks - 1:
        os._exit(17)
    t0 = time.time()
    code = 0
    try:
        comm.allreduce(t, "sum")
    except ab.AstroBurstError as e:
        code = e.code
    dt = time.time() - t0
    code2 = 0
    try:  # the communicator is dead now: fails at once
        comm.allreduce(t, "sum")
    except ab.AstroBurstError as e:
        code2 = e.code
    np.savez(os.path.join(out, f"rank{rank}.npz"), code=np.array([code]), seconds=np.array([dt]), code2=np.array([code2]), seconds2=np.array([time.time() - t0 - dt]))


def scenario_abort(ab, ctx, comm, rank, nranks, out):
    """ab_comm_abort on one rank releases the peers blocked in a collective at once"""
    import torch
    comm.set_timeout_ms(60000)
    t = torch.ones(1000, dtype=torch.float64, device="cuda")
    comm.allreduce(t, "sum")
    t0 = time.time()
    code = 0
    if rank == nranks - 1:
        time.sleep(1.0)
        comm.abort()
    else:
        try:
            comm.allreduce(t, "sum")
        except ab.AstroBurstError as e:
            code = e.code
    np.savez(os.path.join(out, f"rank{rank}.npz"), code=np.array([code]), seconds=np.array([time.time() - t0]))


def band_frames():
    """(reference, 15 targets) of the row-band scenario: shifts up to 9 px and rotations up to 0.6 degrees, so the halos differ
    per rank and per frame; 400 x 640 frames"""
    import math
    import torch
    from astroburst_amd import synth
    rows, cols = 400, 640
    y, x, flux = synth.star_catalog(rows, cols, 420, seed=23)
    g = torch.Generator().manual_seed(99)
    frames, truth_t = [], []
    for k in range(16):
        if k == 0:
            a = b = c = d = tx = ty = 0.0
            a = d = 1.0
        else:
            ang = math.radians(float(torch.rand(1, generator=g)) * 1.2 - 0.6)
            tx, ty = (float(torch.rand(1, generator=g)) * 18.0 - 9.0 for _ in range(2))
            a, b, c, d = math.cos(ang), -math.sin(ang), math.sin(ang), math.cos(ang)
            # rotate about the frame centre
            tx += cols / 2.0 - (a * cols / 2.0 + b * rows / 2.0)
            ty += rows / 2.0 - (c * cols / 2.0 + d * rows / 2.0)
        cat_k = (c * x + d * y + ty, a * x + b * y + tx, flux * 30.0)   # a star at reference (x, y) lands at T_k(x, y)
        frames.append(synth.make_frame(rows, cols, k, cat=cat_k, bad_patch_rate=0.0, cosmic_rate=1e-4))
        truth_t.append((a, b, tx, c, d, ty))
    return frames[0], frames[1:], truth_t[1:]


def scenario_bands(ab, ctx, comm, rank, nranks, out):
    """SURVEY 8e's row-band scheme with every rank INGESTING a band: whole frames only for the targets this rank detects (i mod size)
    and the reference; of every other target just the rows its band of the output reads (ab_shard_source_rows: its rows + the halo the
    exchanged transforms need).  The host slices are cut from the frames BEFORE they reach the device: a rank never holds more."""
    import torch

    def say(what):
        print(f"[bands rank {rank}] {what} at {time.time() % 1000:.1f}", file=sys.stderr, flush=True)
    say("start")
    ref, tgts, _ = band_frames()
    say("frames generated")
    rows, cols = ref.shape
    n = len(tgts)
    mine = [i for i in range(n) if i % nranks == rank]
    # detection is frame-sharded: only this rank's targets (and the reference) are uploaded whole
    dev_targets = [tgts[i].cuda() if i in mine else torch.empty((0, cols), device="cuda") for i in range(n)]
    ref_dev = ref.cuda()
    regs = ctx.register_frames_sharded(comm, ref_dev, dev_targets, num_threads=8)
    transforms = [r.transform for r in regs]
    say("registered")
    # round 6: the same through ab_align_pairs_affine_rowband (own frames whole and warped as they are fitted, the others from
    # their rows + halo after the exchange) -- checked further down against the step-by-step route, pixel for pixel
    row0_, nrows_ = ctx.shard_rows(rows, nranks, rank)
    s0_, sn_ = ctx.shard_source_rows(transforms, rows, cols, rows, cols, nranks, rank)
    one_call_in = [dev_targets[i] if i in mine else tgts[i][s0_:s0_ + sn_].contiguous().cuda() for i in range(n)]
    one_call_out = [torch.empty((nrows_, cols), device="cuda") for _ in range(n)]
    regs1 = ctx.align_pairs_affine_rowband(comm, ref_dev, one_call_in, one_call_out, row0_, target_row0=[0 if i in mine else s0_ for i in range(n)], num_threads=8)
    assert [(r.method, r.transform, r.inliers) for r in regs1] == [(r.method, r.transform, r.inliers) for r in regs], "the one-call registration differs"
    say("registered in one call")
    del dev_targets, one_call_in
    row0, nrows = ctx.shard_rows(rows, nranks, rank)
    s0, sn = ctx.shard_source_rows(transforms, rows, cols, rows, cols, nranks, rank)
    resident = 0
    warped = [ref_dev[row0:row0 + nrows].contiguous()]
    for i in range(n):
        f0, fn = ctx.warp_source_rows(transforms[i], rows, cols, cols, row0, nrows)
        assert fn == 0 or (f0 >= s0 and f0 + fn <= s0 + sn)
        band_host = tgts[i][s0:s0 + sn].contiguous()          # what this rank is GIVEN of target i
        band_dev = band_host.cuda()
        resident += band_dev.numel() * 4
        o = torch.empty((nrows, cols), device="cuda")
        ctx.warp_image_rows_from_band(band_dev, s0, rows, transforms[i], rows, row0, o)
        assert torch.equal(torch.nan_to_num(o, nan=-1.0), torch.nan_to_num(one_call_out[i], nan=-1.0)), f"frame {i}: the one-call band differs"
        warped.append(o)
    # (the frames are bands already: the band-local stack + the rejected count summed over the ranks)
    say("warped")
    out_band, rej_local = ctx.stack_sigma_clip(warped, 3.0, 3.0, 5) if nrows > 0 else (torch.empty((0, cols), device="cuda"), 0)
    say("stacked")
    t = torch.tensor([rej_local], dtype=torch.int64, device="cuda")
    comm.allreduce(t, "sum")
    full = torch.zeros((rows, cols), device="cuda")
    ctx.allgather_rows(comm, out_band, full)
    say("gathered")
    # a band that is too short is refused loudly, naming the rows: target 0 without the last row its band of the output reads
    code = 0
    f0, fn = ctx.warp_source_rows(transforms[0], rows, cols, cols, row0, nrows)
    if nrows > 0 and fn > 1:
        try:
            ctx.warp_image_rows_from_band(tgts[0][f0:f0 + fn - 1].contiguous().cuda(), f0, rows, transforms[0], rows, row0, torch.empty((nrows, cols), device="cuda"))
        except ab.AstroBurstError as e:
            code = e.code
            assert "read source rows" in e.message
    np.savez(os.path.join(out, f"rank{rank}.npz"), full=full.cpu().numpy(), rej=np.array([int(t.item())]), transforms=np.array(transforms),
             src=np.array([s0, sn, row0, nrows]), resident=np.array([resident]), short_band_code=np.array([code]))


def scenario_sum(ab, ctx, comm, rank, nranks, out):
    """one all-reduce: proves that every rank of the job sits in the same segment"""
    import torch
    t = torch.full((1000,), float(rank + 1), dtype=torch.float64, device="cuda")
    comm.allreduce(t, "sum")
    np.savez(os.path.join(out, f"rank{rank}.npz"), total=np.array([float(t[0].item())]))


def main():
    name, nranks, rank, scenario, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import torch
    torch.set_num_threads(2)   # eight ranks generate their inputs on the host at once: eight full-width thread pools starve each other
    import astroburst_amd as ab
    ctx = ab.Context(0)
    ctx.use_torch_stream()
    comm = ab.Comm.host(ctx, name, nranks, rank)
    assert comm.is_host and (comm.rank, comm.size) == (rank, nranks)
    {"all": scenario_all, "fail": scenario_fail, "die": scenario_die, "abort": scenario_abort, "sum": scenario_sum, "bands": scenario_bands}[scenario](ab, ctx, comm, rank, nranks, out)
    comm.close()
    ctx.close()


if __name__ == "__main__":
    main()
