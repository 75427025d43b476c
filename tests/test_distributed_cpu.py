"""world_size-2 gloo tests of the multi-rank paths (bench.py --gpus N): frame shards and exact row bands.

No GPU here, so the per-rank partial is computed by the CPU oracle (the same function that checks
the HIP partial kernel in tests/test_gpu_stack.py::test_partial_two_level); what this test covers is
the sharding, the all-reduce of (sum, count, rejected) and the divide -- i.e. that N ranks produce
exactly the two-level estimate of the union of their shards."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from astroburst_amd import synth
    from astroburst_amd.distributed import shard_frames, sharded_stack
    from oracle import pyoracle

    frames = synth.make_stack(n_frames, 48, 64)           # every rank can regenerate any frame (deterministic)
    mine = [frames[k] for k in shard_frames(n_frames, world, rank)]

    def partial_fn(fr):
        s, c, rej = pyoracle.stack_partial([f.numpy() for f in fr])
        return torch.from_numpy(s), torch.from_numpy(c.astype(np.int32)), rej

    def finalize_fn(s, c):
        s, c = s.numpy(), c.numpy()
        return np.where(c > 0, (s / np.maximum(c, 1)).astype(np.float32), np.float32(0))

    img, rej = sharded_stack(mine, partial_fn, finalize_fn)
    if rank == 0:
        np.save(out_path, np.concatenate([img.ravel(), np.array([rej], np.float32)]))
    dist.destroy_process_group()


def test_two_rank_sharded_stack_equals_two_level_oracle(tmp_path):
    from astroburst_amd import synth
    from astroburst_amd.distributed import shard_frames
    from oracle import pyoracle

    n_frames, world = 13, 2                                # uneven shards: 7 + 6
    out = str(tmp_path / "r0.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_frames, out), nprocs=world, join=True)
    got = np.load(out)
    img, rej = got[:-1].reshape(48, 64), int(got[-1])

    frames = [f.numpy() for f in synth.make_stack(n_frames, 48, 64)]
    tot_s, tot_c, tot_r = 0.0, 0, 0
    for r in range(world):
        idx = shard_frames(n_frames, world, r)
        s, c, rj = pyoracle.stack_partial([frames[k] for k in idx])
        tot_s, tot_c, tot_r = tot_s + s, tot_c + c.astype(np.int64), tot_r + rj
    want = np.where(tot_c > 0, (tot_s / np.maximum(tot_c, 1)).astype(np.float32), np.float32(0))
    assert np.array_equal(img, want)
    assert rej == tot_r
    # and the two-level estimate stays close to the reference's single-level stack
    single, _ = pyoracle.stack_images(frames)
    assert np.nanmax(np.abs(img - single) / np.maximum(np.abs(single), 1e-6)) < 5e-2


def test_shard_frames_partition():
    from astroburst_amd.distributed import shard_frames
    for n in (1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            if w > n:
                continue
            got = [k for r in range(w) for k in shard_frames(n, w, r)]
            assert got == list(range(n))
            sizes = [len(shard_frames(n, w, r)) for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


# ---- row bands: the exact single-level estimator (SURVEY 8e default; csrc/sharded.hip on the GPU) ------------------------
def _rowband_worker(rank, world, port, n_frames, rows, cols, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import stats_protocol as sp
    from astroburst_amd import synth
    from astroburst_amd.distributed import rowband_stack
    from oracle import pyoracle

    frames = [f.numpy() for f in synth.make_stack(n_frames, rows, cols)]   # every rank holds every frame (row bands of all frames)

    def stack_rows_fn(fr, row0, nrows):
        if nrows == 0:
            return torch.empty((0, cols), dtype=torch.float32), 0
        img, rej = pyoracle.stack_images([f[row0:row0 + nrows] for f in fr])
        return torch.from_numpy(img), rej

    band, rej, (row0, nrows) = rowband_stack(frames, stack_rows_fn, rows, cols, gather=False)
    full, rej2, _ = rowband_stack(frames, stack_rows_fn, rows, cols, gather=True)
    assert rej == rej2

    def ar_sum(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    def ar_max(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.numpy()

    st = sp.stats_hist_sharded(band.numpy(), ar_sum, ar_max)   # statistics of the WHOLE image from this rank's band
    if rank == 0:
        np.savez(out_path, full=full.numpy(), rej=rej, **{k: np.float64(v) for k, v in st.items()})
    dist.destroy_process_group()


def _check_rowband(tmp_path, world, n_frames, rows, cols):
    from astroburst_amd import synth
    from oracle import pyoracle

    out = str(tmp_path / f"rb{world}.npz")
    mp.spawn(_rowband_worker, args=(world, _free_port(), n_frames, rows, cols, out), nprocs=world, join=True)
    got = np.load(out)
    frames = [f.numpy() for f in synth.make_stack(n_frames, rows, cols)]
    want, want_rej = pyoracle.stack_images(frames)             # the reference's single-level estimator, one process
    assert np.array_equal(got["full"], want, equal_nan=True)   # bit for bit (unlike the two-level frame shards)
    assert int(got["rej"]) == want_rej
    ws = pyoracle.compute_image_stats(want, path="hist")
    assert int(got["valid_count"]) == ws.valid_count
    for k in ("min", "max", "median", "mad", "sigma"):         # from integer partials only: exact
        assert float(got[k]) == getattr(ws, k), k
    assert abs(float(got["mean"]) - ws.mean) <= 1e-12 * abs(ws.mean)   # the one f64 sum (reduction order)


def test_two_rank_rowband_stack_and_stats_equal_single_level_oracle(tmp_path):
    _check_rowband(tmp_path, 2, 9, 49, 64)     # 25 + 24 rows


def test_three_rank_rowband_with_a_short_band(tmp_path):
    _check_rowband(tmp_path, 3, 5, 10, 40)     # 4 + 4 + 2 rows


def _empty_rank_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from astroburst_amd.distributed import sharded_stack
    frames = [torch.zeros((4, 4))] if rank == 0 else []       # rank 1 has nothing: every rank must raise, none may hang
    try:
        sharded_stack(frames, lambda fr: (_ for _ in ()).throw(AssertionError("no data collective may start")), None)
        verdict = "returned"
    except ValueError:
        verdict = "raised"
    open(os.path.join(out_dir, f"v{rank}"), "w").write(verdict)
    dist.destroy_process_group()


def test_mis_sharded_job_raises_on_every_rank(tmp_path):
    mp.spawn(_empty_rank_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert [open(tmp_path / f"v{r}").read() for r in range(2)] == ["raised", "raised"]


def test_shard_partitions_match_the_library():
    """the Python protocol and the C ABI (ab_shard_rows / ab_shard_frames: host arithmetic, no GPU) agree"""
    import ctypes as C
    from astroburst_amd import _lib
    from astroburst_amd.distributed import shard_frames, shard_rows
    L = _lib.lib()
    for n in (1, 7, 10, 49, 4096, 13759):
        for w in (1, 2, 3, 4, 8):
            cover = []
            for r in range(w):
                r0, nr = C.c_int64(), C.c_int64()
                assert L.ab_shard_rows(n, w, r, C.byref(r0), C.byref(nr)) == 0
                rr = shard_rows(n, w, r)
                assert (rr.start if len(rr) else min(n, -(-n // w) * r), len(rr)) == (r0.value, nr.value)
                cover += list(rr)
                f0, nf = C.c_size_t(), C.c_size_t()
                assert L.ab_shard_frames(n, w, r, C.byref(f0), C.byref(nf)) == 0
                fr = shard_frames(n, w, r)
                assert (fr.start, len(fr)) == (f0.value, nf.value)
            assert cover == list(range(n))
