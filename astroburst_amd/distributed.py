"""Sharding logic of the hot path across ranks, with the compute and the collectives injected.

On the MI355X the compute is libastroburst_hip.so and the collectives are RCCL inside the library
(`Context.stack_sigma_clip_sharded`, `Context.stack_sigma_clip_rowband`, `Context.compute_image_stats_sharded`:
csrc/sharded.hip, csrc/comm.hip) -- a Rust host needs nothing from this module.  What lives here is the same
partition / exchange protocol written against `torch.distributed`, so that it can be exercised on CPU ranks
(`gloo`, tests/test_distributed_cpu.py) with the oracle standing in for the kernels:

* row bands (exact): rank r stacks rows [row0, row0 + nrows) of ALL frames (the reference's per-pixel loop,
  core/stacking/combine.rs:160-182, restricted to a band); exchanged: one u64 (rejected samples) and, for the
  statistics of the result, min / max, counts and 65 536-bin histograms -- integers only, so N ranks reproduce the
  single-process result bit for bit.
* frame shards (BASELINE configs[3]): every rank clips ITS frames to per-pixel (f64 sum, u32 count); the partials
  are summed with one all-reduce each and divided.  A two-level estimator: it equals the reference's single-level
  `stack_images` only where no sample is rejected (or the rejections coincide); checker: oracle.stack_partial.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_frames(n_frames: int, world: int, rank: int) -> range:
    """Contiguous, balanced frame ranges: rank r gets frames [lo, hi)  (== ab_shard_frames)."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return range(lo, hi)


def shard_rows(rows: int, world: int, rank: int) -> range:
    """ceil(rows / world) rows per rank, trailing bands shorter or empty  (== ab_shard_rows)."""
    per = -(-rows // world)
    return range(min(rows, per * rank), min(rows, per * (rank + 1)))


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_initialized() else 0


def sharded_stack(local_frames, partial_fn, finalize_fn, group=None):
    """Frame-sharded two-level stack.  Returns (image, total_rejected).

    `partial_fn(frames) -> (sum f64, count i32, rejected)`, `finalize_fn(sum, count) -> image`.  Every rank must hold at
    least one frame: the shard sizes are agreed on BEFORE any data collective, so a mis-sharded job raises on every
    rank instead of leaving the others waiting in an all-reduce."""
    world = _world(group)
    if world > 1:
        n_min = torch.tensor([len(local_frames)], dtype=torch.int64)
        dist.all_reduce(n_min, op=dist.ReduceOp.MIN, group=group)
        if int(n_min.item()) == 0:
            raise ValueError("every rank needs at least one frame (world size > frame count)")
    elif len(local_frames) == 0:
        raise ValueError("every rank needs at least one frame (world size > frame count)")
    s, c, rej = partial_fn(local_frames)
    if world > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
        c_work = c if c.dtype != torch.uint32 else c.to(torch.int32)
        dist.all_reduce(c_work, op=dist.ReduceOp.SUM, group=group)
        c = c_work
        r = torch.tensor([rej if rej is not None else 0], dtype=torch.int64, device=s.device)
        dist.all_reduce(r, op=dist.ReduceOp.SUM, group=group)
        rej = int(r.item())
    return finalize_fn(s, c), rej


def rowband_stack(frames, stack_rows_fn, rows: int, cols: int, group=None, gather: bool = True):
    """Row-band exact stack.  `stack_rows_fn(frames, row0, nrows) -> (band f32 [nrows, cols] torch tensor, rejected)`.

    Returns (image or this rank's band, total_rejected, (row0, nrows)).  With gather=True every rank ends with the whole
    image (bands all-gathered, padded to the common band height as RCCL / gloo all-gather needs equal blocks)."""
    world, rank = _world(group), _rank(group)
    rr = shard_rows(rows, world, rank)
    row0, nrows = rr.start, len(rr)
    band, rej = stack_rows_fn(frames, row0, nrows)
    if world > 1:
        r = torch.tensor([rej], dtype=torch.int64, device=band.device)
        dist.all_reduce(r, op=dist.ReduceOp.SUM, group=group)
        rej = int(r.item())
    if not gather:
        return band, rej, (row0, nrows)
    if world == 1:
        return band, rej, (row0, nrows)
    per = -(-rows // world)
    mine = torch.zeros((per, cols), dtype=band.dtype, device=band.device)
    mine[:nrows] = band
    blocks = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(blocks, mine, group=group)
    full = torch.cat(blocks, dim=0)[:rows]
    return full, rej, (row0, nrows)


def allreduce_sum_(t, group=None):
    if _world(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def allreduce_max_(t, group=None):
    if _world(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return t
