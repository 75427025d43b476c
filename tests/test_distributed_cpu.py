"""world_size-2 gloo test of the frame-sharded stacking path (bench.py --gpus N).

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
