"""Flood complex across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-device.  Two decompositions are provided.

``mode="simplices"`` (default): every rank holds the whole cloud (12 MB per million 3D points - nothing
against 288 GB) and sweeps every W-th simplex of the sorted simplex list; the per-face filtration values
(+inf where another rank is responsible) are combined with one ``all_reduce(MIN)``.  On the default cell-sweep path
(2-D / 3-D, float32) the buffer is ONE word per distinct face of the complex - the simplices of a rank that share a
triangle, an edge or a vertex share its running maximum, as on a single GPU (``core.shared_face_slots``,
``core.shard_slot_fill``) -, otherwise the (S, F) matrix.  The culled sweeps do work proportional to the number of
SAMPLES, not points, so this is the decomposition that scales; the exchanged buffer is about a hundred KB.

Above 3 dimensions, where a dimension pass runs through the sorted-sample sweep (tiles of 64 spatially consecutive
samples of ALL simplices, ``csrc/flood_sorted.hip``), ``mode="simplices"`` shards the TILES of the sorted order
instead of the simplices: every rank sorts all samples (the same order everywhere), sweeps a contiguous W-th of the
tiles - exactly tiles of the unsharded sweep, in one region of space; every W-th simplex would thin the samples W times and widen the tiles
by W^(1/dim) - and holds, per face, the maximum over ITS samples; the ranks' (S, F) matrices combine with MAX (the
hook's MIN applied to the negated non-negative values).

``mode="points"``: the path also shards over the point cloud because
``min_x |p - x|`` is associative: every rank holds all simplices (tiny) and ANY subset of the
points, computes the per-sample minimum squared distance against its subset, and one
``all_reduce(MIN)`` on the (S, R) buffer of squared-distance bit patterns gives the global minimum.
The reduction has to happen on the per-sample minima, BEFORE the per-face maximum
(``min_g max_r`` is not ``max_r min_g``), which is what the ``reduce_hook`` of
``flood_complex`` provides.

``mode="blocks"``: as ``"simplices"``, but rank r takes a CONTIGUOUS block of the simplex queue (the simplices are
ordered along the widest axis of the cloud: a slab of space) and builds its index over the part of the cloud inside
the bounding balls of its block only (``core.block_subcloud``) - with landmarks that are cloud points (every
``generate_landmarks`` result; the precondition of the reference's own GPU path, ``core.py:156-172``) every witness
of a simplex lies in its ball.  The index build, which every rank of the other two modes repeats over the whole
cloud, shrinks with the share; the values are produced whole by one rank each and ``all_reduce(MIN)`` on the (S, F)
matrix completes them - BASELINE.json's "the point cloud shards ... all-reduce(min) on the per-simplex filtration
values".  Blocks are not work-balanced (a slab through the dense core of a Gaussian holds the expensive simplices).

Shards are interleaved (rank r takes sorted rows r, r+W, r+2W, ...): every rank then sees 1/W of
each simplex's candidates, so the heavy-tailed per-simplex work balances without any planning.
"""

from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from .core import flood_complex

__all__ = ["shard_points", "min_reduce_hook", "global_widest_axis", "sync_cpu_rng", "flood_complex_sharded"]


def shard_points(points: torch.Tensor, rank: int, world_size: int) -> torch.Tensor:
    """Rank ``rank``'s interleaved share of ``points`` (any partition gives the same result)."""
    return points[rank::world_size].contiguous()


def min_reduce_hook(group: Optional[dist.ProcessGroup] = None, always: bool = False):
    """``reduce_hook`` for ``flood_complex``: in-place ``all_reduce(MIN)`` over the process group.

    ``always``: issue the collective also on a group of ONE rank (where it changes nothing) - the way the RCCL
    path is exercised on a single GPU: ``tests/test_distributed_gpu.py::test_rccl_world_size_one``.

    On ROCm tensors the buffer holds bit patterns of non-negative squared distances, for which integer order ==
    numeric order: int32 words of float32 values (+inf = 0x7f800000) for float32 inputs, int64 words of float64
    values (+inf = 0x7ff0000000000000) for float64 inputs; on CPU it is the float distance matrix.  A hook that
    wants numbers must view the buffer by its dtype (``buf.view(torch.float32 if buf.dtype == torch.int32 else
    torch.float64)``); MIN needs no view.
    """

    def hook(buf: torch.Tensor) -> None:
        if dist.is_available() and dist.is_initialized() and (always or dist.get_world_size(group) > 1):
            dist.all_reduce(buf, op=dist.ReduceOp.MIN, group=group)

    # a real collective: flood_complex first lets the ranks compare the shape of what they are about to reduce (a
    # six-word MIN through this same hook) and raises on every rank instead of hanging in a mismatched all-reduce
    hook.checks_ranks = True
    return hook


def global_widest_axis(points_shard: torch.Tensor, group: Optional[dist.ProcessGroup] = None, always: bool = False) -> int:
    """Axis of largest extent of the WHOLE cloud (what ``core.py:140-142`` computes on one device),
    from per-shard extrema combined with two tiny all-reduces, so that every rank sorts its simplices
    the same way."""
    lo = points_shard.min(dim=0).values.to(torch.float32)
    hi = points_shard.max(dim=0).values.to(torch.float32)
    if dist.is_available() and dist.is_initialized() and (always or dist.get_world_size(group) > 1):
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return int(torch.argmax(hi - lo).item())


def sync_cpu_rng(device: torch.device, group: Optional[dist.ProcessGroup] = None, src: int = 0, always: bool = False) -> None:
    """Give every rank rank ``src``'s global CPU generator state.

    ``generate_uniform_weights`` (``core.py:405-427`` of the reference) draws the random barycentric
    weights from the global CPU generator.  Column r of the (S, R) buffer that ``all_reduce(MIN)`` combines
    must be the same sample point on every rank, and in ``mode="simplices"`` every rank's simplices must be
    sampled with the weights an unsharded run would use, so the generator state is broadcast (about 5 KB)
    before the weights are drawn: the result equals the unsharded result under rank ``src``'s seed,
    whatever the other ranks' generators held."""
    if not (dist.is_available() and dist.is_initialized() and (always or dist.get_world_size(group) > 1)):
        return
    state = torch.get_rng_state()
    buf = state.to(device) if device.type == "cuda" else state.clone()
    dist.broadcast(buf, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    torch.set_rng_state(buf.cpu())


def flood_complex_sharded(points: torch.Tensor, landmarks: torch.Tensor, *args, mode: str = "simplices",
                          group: Optional[dist.ProcessGroup] = None, always_reduce: bool = False, **kwargs):
    """``flood_complex`` over all ranks of ``group``; every rank returns the full result.

    ``mode="simplices"``: ``points`` is the FULL cloud on every rank, simplices are interleaved over the
    ranks.  ``mode="blocks"``: the FULL cloud on every rank, contiguous blocks of simplices, each rank indexes only the
    sub-cloud its block can see (landmarks must be points of the cloud).  ``mode="points"``: ``points`` is this rank's shard of the cloud (``shard_points``).  The
    landmark tensor must be identical on every rank.  Other arguments as ``flood_complex``.

    With ``num_rand`` the sample weights come from the global CPU generator; rank 0's generator state is
    broadcast first (``sync_cpu_rng``), so seeding rank 0 is enough and the result equals the unsharded
    one under that seed.  ``always_reduce``: run the collectives on a one-rank group too (``min_reduce_hook``)."""
    if not isinstance(landmarks, torch.Tensor):
        raise TypeError("flood_complex_sharded needs explicit landmark coordinates (identical on every "
                        "rank); run generate_landmarks on the full cloud first")
    num_rand = kwargs.get("num_rand", args[2] if len(args) > 2 else None)
    if num_rand is not None:
        sync_cpu_rng(points.device, group, always=always_reduce)
    if mode == "points":
        axis = global_widest_axis(points, group, always_reduce)
        return flood_complex(points, landmarks, *args, reduce_hook=min_reduce_hook(group, always_reduce),
                             sort_axis=axis, **kwargs)
    if mode not in ("simplices", "blocks"):
        raise ValueError("mode must be 'simplices', 'blocks' or 'points'")
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    return flood_complex(points, landmarks, *args, simplex_shard=(rank, world), shard_blocks=(mode == "blocks"),
                         face_reduce_hook=min_reduce_hook(group, always_reduce), **kwargs)
