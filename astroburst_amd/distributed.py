"""Frame-sharded stacking across ranks (BASELINE config C4).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).
Every rank clips ITS OWN frames to a per-pixel partial (f64 sum of survivors, u32 count of
survivors); the partials are summed with one all-reduce each and divided.  The only data-path
collective is that all-reduce: 12 bytes per pixel (8 + 4), ring-bound by the per-link xGMI
bandwidth, independent of the number of frames.

This is a two-level estimator: clipping is decided inside a shard, so it equals the reference's
single-level `stack_images` only when no sample is rejected (or the rejections coincide).  Its CPU
checker is oracle.stack_partial on the same shards.

`partial_fn(frames) -> (sum, count, rejected)` and `finalize_fn(sum, count) -> image` are injected so
the sharding/reduction logic is testable without a GPU: on the MI355X they are
Context.stack_partial / Context.stack_finalize_partial (HIP kernels); the gloo test passes the oracle.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_frames(n_frames: int, world: int, rank: int) -> range:
    """Contiguous, balanced frame ranges: rank r gets frames [lo, hi)."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return range(lo, hi)


def sharded_stack(local_frames, partial_fn, finalize_fn, group=None):
    """Returns (image, total_rejected).  local_frames: this rank's frames (may be empty)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if len(local_frames) > 0:
        s, c, rej = partial_fn(local_frames)
    else:
        raise ValueError("every rank needs at least one frame (world size > frame count)")
    if world > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
        c_work = c if c.dtype != torch.uint32 else c.to(torch.int32)
        dist.all_reduce(c_work, op=dist.ReduceOp.SUM, group=group)
        c = c_work
        r = torch.tensor([rej if rej is not None else 0], dtype=torch.int64, device=s.device)
        dist.all_reduce(r, op=dist.ReduceOp.SUM, group=group)
        rej = int(r.item())
    return finalize_fn(s, c), rej
