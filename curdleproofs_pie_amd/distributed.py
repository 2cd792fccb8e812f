"""Multi-GPU MSM: one process per GPU, one tiny exchange step.

A single large MSM is partitioned across the ranks of one node either
  * by WINDOW  ("windows", what BASELINE.json's north_star names): rank g owns the signed-digit windows
    w = g (mod world); every rank holds all n points/scalars and returns  sum_{w in g} 2^(c w) S_w ; or
  * by POINT   ("points"): rank g owns points [g*n/world, (g+1)*n/world) with all windows.
Either way the partials are ONE G1 point per rank.  RCCL has no elliptic-curve reduction operator, so the
"all-reduce of partial G1 sums" is an all-gather of `world` 144-byte blobs (over xGMI, latency-bound: ~1 KB)
followed by world-1 host G1 additions on every rank -- every rank ends with the same, bit-exact result
(G1 addition is commutative/associative and the final encoding is canonical).

torch.distributed is used only as the transport (backend "nccl" == RCCL on ROCm; "gloo" in CPU tests);
the MSM itself never touches torch.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

from . import _native as N


def sum_blobs(blobs: List[bytes]) -> bytes:
    """Host G1 sum of point blobs (the reduction operator of the 'all-reduce')."""
    acc = ctypes.create_string_buffer(N.POINT_BYTES)
    N.cg1_identity(acc)
    for b in blobs:
        N.cg1_add(acc, acc.raw, bytes(b))
    return acc.raw


def all_reduce_g1(partial_blob: bytes, group=None, device: Optional[str] = None) -> bytes:
    """All-gather every rank's partial G1 point and add them; returns the same blob on all ranks."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bytes(partial_blob)
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = device or ("cuda" if backend == "nccl" else "cpu")
    mine = torch.frombuffer(bytearray(partial_blob), dtype=torch.uint8).to(dev)
    gathered = [torch.empty(N.POINT_BYTES, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    return sum_blobs([bytes(t.cpu().numpy().tobytes()) for t in gathered])


def sharded_msm(ctx: "N.Context", d_points, d_scalars, n: int, rank: int, world: int, window_c: int = 16,
                mode: str = "windows", group=None) -> bytes:
    """This rank's share of one MSM, then the G1 all-reduce.

    mode "windows": d_points/d_scalars hold ALL n terms on every rank.
    mode "points" : d_points/d_scalars hold only this rank's shard of n terms.
    """
    if mode == "windows":
        part = ctx.msm_device(d_points, d_scalars, n, window_c=window_c, shard_rank=rank, shard_world=world)
    elif mode == "points":
        part = ctx.msm_device(d_points, d_scalars, n, window_c=window_c)
    else:
        raise ValueError(f"unknown shard mode {mode!r}")
    return all_reduce_g1(part, group=group)
