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


_bufs = {}


def all_reduce_g1(partial_blob: bytes, group=None, device: Optional[str] = None) -> bytes:
    """All-gather every rank's partial G1 point and add them; returns the same blob on all ranks.

    One collective per call: a single all-gather of 144 bytes per rank into a cached device buffer, one D2H copy,
    world-1 host additions."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bytes(partial_blob)
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = device or ("cuda" if backend == "nccl" else "cpu")
    key = (dev, world)
    if key not in _bufs:
        _bufs[key] = (torch.empty(N.POINT_BYTES, dtype=torch.uint8, device=dev),
                      torch.empty(world * N.POINT_BYTES, dtype=torch.uint8, device=dev))
    mine, gathered = _bufs[key]
    mine.copy_(torch.frombuffer(bytearray(partial_blob), dtype=torch.uint8))
    try:
        dist.all_gather_into_tensor(gathered, mine, group=group)
        raw = gathered.cpu().numpy().tobytes()
    except (RuntimeError, NotImplementedError):      # transports without the flat form
        lst = [torch.empty(N.POINT_BYTES, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(lst, mine, group=group)
        raw = b"".join(t.cpu().numpy().tobytes() for t in lst)
    return sum_blobs([raw[i * N.POINT_BYTES:(i + 1) * N.POINT_BYTES] for i in range(world)])


def shard_layout(rank: int, world: int, mode: str = "hybrid", window_groups: int = 2):
    """-> (window_rank, window_groups W, point_rank, point_groups P) with W * P == world.

    "windows": W = world (every rank holds ALL points, owns the windows w = rank mod world) -- replicates the
               per-point work (k_prepare_points, digit extraction) world times: 4.25 ms per rank at world = 8;
    "points":  P = world (every rank holds 1/world of the points, all windows);
    "hybrid":  W = window_groups (default 2) window-bucket groups x P = world / W point groups: window buckets are still
               sharded across GPUs, but a rank only touches 1/P of the points (3.47 ms per rank at world = 8 on MI355X,
               tools/gpu_shard_emulation.py).  Falls back to "windows" when world is not a multiple of window_groups."""
    if mode == "windows" or (mode == "hybrid" and (world % window_groups or world < window_groups)):
        return rank, world, 0, 1
    if mode == "points":
        return 0, 1, rank, world
    if mode != "hybrid":
        raise ValueError(f"unknown shard mode {mode!r}")
    W = window_groups
    return rank % W, W, rank // W, world // W


def sharded_msm(ctx: "N.Context", d_points, d_scalars, n: int, rank: int, world: int, window_c: int = 16,
                mode: str = "windows", group=None) -> bytes:
    """This rank's share of one MSM, then the G1 all-reduce.

    mode "windows": d_points/d_scalars hold ALL n terms on every rank.
    mode "points" : d_points/d_scalars hold only this rank's shard of n terms.
    mode "hybrid" : d_points/d_scalars hold the n terms of this rank's point group (see shard_layout).
    """
    if mode == "windows":
        part = ctx.msm_device(d_points, d_scalars, n, window_c=window_c, shard_rank=rank, shard_world=world)
    elif mode == "points":
        part = ctx.msm_device(d_points, d_scalars, n, window_c=window_c)
    elif mode == "hybrid":              # d_points/d_scalars hold this rank's POINT GROUP (n terms); windows split inside the group
        wr, W, _, _ = shard_layout(rank, world, "hybrid")
        part = ctx.msm_device(d_points, d_scalars, n, window_c=window_c, shard_rank=wr, shard_world=W)
    else:
        raise ValueError(f"unknown shard mode {mode!r}")
    return all_reduce_g1(part, group=group)


def sharded_msm_batch(jobs_affine, rank: int, world: int, group=None, compute=None) -> List[bytes]:
    """Many INDEPENDENT MSMs (e.g. one final accumulator MSM per proof; BASELINE config 5: 16384 proofs over
    8 GPUs) sharded job-per-GPU: rank g computes jobs g, g+world, ... with the regime-B batched kernels, then the
    per-job result blobs are all-gathered so every rank holds all results.  Embarrassingly parallel: the only
    collective is that final gather of 144 bytes per job (no data-path exchange).

    jobs_affine: list of (points_affine96: bytes, scalars32: bytes, n: int).
    compute: callable(list_of_jobs) -> list of blobs for this rank's share; defaults to the GPU batched path
             (curdleproofs_pie_amd has no CPU path; the parameter exists so the gloo tests can exercise the
             sharding/gather logic on a CPU-only box with the host operators).
    """
    mine = list(range(rank, len(jobs_affine), world))
    my_jobs = [jobs_affine[i] for i in mine]
    if compute is None:
        def compute(js):
            if not js:
                return []
            offs = [0]
            for _, _, n in js:
                offs.append(offs[-1] + n)
            return N.default_context().msm_batched_host(b"".join(p for p, _, _ in js), b"".join(s for _, s, _ in js), offs)
    my_blobs = compute(my_jobs)
    assert len(my_blobs) == len(my_jobs)
    import torch.distributed as dist

    if world == 1 or not dist.is_initialized():
        out = [None] * len(jobs_affine)
        for i, b in zip(mine, my_blobs):
            out[i] = bytes(b)
        return out
    import torch

    # fixed-size tensor gather: rank g owns jobs g, g + world, ... (known to every rank, so no indices travel), each rank
    # sends ceil(jobs / world) slots of 144 bytes, unused slots zero
    width = (len(jobs_affine) + world - 1) // world
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    local = torch.zeros(width * N.POINT_BYTES, dtype=torch.uint8)
    if my_blobs:
        for b in my_blobs:
            assert len(b) == N.POINT_BYTES
        local[: len(my_blobs) * N.POINT_BYTES] = torch.frombuffer(bytearray(b"".join(bytes(b) for b in my_blobs)), dtype=torch.uint8)
    local = local.to(dev)
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local, group=group)
    out = [None] * len(jobs_affine)
    for g, t in enumerate(gathered):
        raw = t.cpu().numpy().tobytes()
        for k, i in enumerate(range(g, len(jobs_affine), world)):
            out[i] = raw[k * N.POINT_BYTES:(k + 1) * N.POINT_BYTES]
    return out


def sharded_verify(verifier, instances: bytes, proofs: bytes, n: int, rank: int, world: int, group=None, mode: str = "merged",
                   verify=None) -> List[int]:
    """BASELINE config 5 (16 384 Whisk shuffle verifications over 8 GPUs): proof-per-GPU sharding.  Rank g verifies the
    contiguous slice [g*n/world, (g+1)*n/world) of the packed batch with its own `ShuffleBatchVerifier` (one merged MSM
    per rank); the per-proof status codes (0 = valid) are all-gathered so every rank returns the full list.
    No data-path collective: the only exchange is that gather of one small integer per proof.

    verify: callable(instances_slice, proofs_slice, m) -> list of m status ints; defaults to verifier.verify_packed
            (the parameter exists so the gloo tests can exercise the slicing / gather logic on a CPU-only box).
    """
    crs = verifier.crs
    inst_b, proof_b = 4 * crs.ell * 48, crs.proof_bytes
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    if verify is None:
        verify = lambda a, b, m: verifier.verify_packed(a, b, m, mode=mode)
    mine = list(verify(instances[lo * inst_b: hi * inst_b], proofs[lo * proof_b: hi * proof_b], hi - lo)) if hi > lo else []
    assert len(mine) == hi - lo
    import torch.distributed as dist

    if world == 1 or not dist.is_initialized():
        return mine
    import torch

    # fixed-size tensor gather (n/world differs by at most one between ranks: pad to the maximum)
    width = (n + world - 1) // world
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    local = torch.full((width,), -1, dtype=torch.int32, device=dev)
    if mine:
        local[: len(mine)] = torch.tensor(mine, dtype=torch.int32, device=dev)
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local, group=group)
    out: List[int] = []
    for g, t in enumerate(gathered):
        cnt = (n * (g + 1)) // world - (n * g) // world
        out.extend(int(x) for x in t[:cnt].cpu().tolist())
    return out
