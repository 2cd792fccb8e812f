#!/usr/bin/env python3
"""Emulated per-rank time of ONE MSM of N x 2^20 terms on N GPUs for every (window groups W) x (point groups P) = N split
(rank 0's share timed on this GPU): windows-only = (N, 1), points-only = (1, N)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N
ctx = N.Context(0)
nmax = 8 << 20
g = ctypes.create_string_buffer(144); N.cg1_generator(g)
a = ctypes.create_string_buffer(96); N.cg1_to_affine96(a, g.raw)
dg, dk, dp, ds = ctx.alloc(96), ctx.alloc(32 * nmax), ctx.alloc(96 * nmax), ctx.alloc(32 * nmax)
dg.upload(a.raw); ctx.gen_scalars_device(dk, nmax, 5); ctx.batch_mul_device(dg, 1, dk, dp, nmax); ctx.gen_scalars_device(ds, nmax, 6)
def med(f, reps=7):
    f(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3
base = med(lambda: ctx.msm_device(dp, ds, 1 << 20, window_c=16))
print(f"single GPU, 2^20 terms: {base:.3f} ms")
for world in (2, 4, 8):
    for W in (1, 2, 4, 8, 16):
        if world % W: continue
        P = world // W
        n_rank = (world << 20) // P
        w = med(lambda: ctx.msm_device(dp, ds, n_rank, window_c=16, shard_rank=0, shard_world=W))
        tm = ctx.timings()
        print(f"N={world}: windows x{W} . points x{P}: rank time {w:.3f} ms -> {world * base / w:.2f}x of one GPU | " +
              " ".join(f"{k}={v:.3f}" for k, v in tm.items() if k in ("prepare", "sort_count", "sort_scatter", "chunks", "accumulate", "seg_reduce", "bit_tree", "host_tail")), flush=True)
