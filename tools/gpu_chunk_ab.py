#!/usr/bin/env python3
"""Minimum chunk length ("chunk_len") at mid sizes, after the round-4 changes to the reduction tail."""
import ctypes, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curdleproofs_pie_amd import _native as N  # noqa: E402
ctx = N.Context(0)
ctx.set_param("profile", 2)
nmax = 1 << 20
dk, dp, ds, dg = ctx.alloc(32 * nmax), ctx.alloc(96 * nmax), ctx.alloc(32 * nmax), ctx.alloc(96)
g = ctypes.create_string_buffer(N.POINT_BYTES); N.cg1_generator(g)
a = ctypes.create_string_buffer(96); N.cg1_to_affine96(a, g.raw)
dg.upload(a.raw)
ctx.gen_scalars_device(dk, nmax, 1)
ctx.batch_mul_device(dg, 1, dk, dp, nmax)
ctx.gen_scalars_device(ds, nmax, 2)
for logn in (17, 18, 19, 20):
    n = 1 << logn
    ref = None
    for L in (8, 20, 24, 32, 40, 48, 80):
        ctx.set_param("chunk_len", L)
        for _ in range(4):
            out = ctx.msm_device(dp, ds, n)
        ref = ref or out
        assert N.cg1_eq(out, ref) == 1
        ws, ph = [], {}
        for _ in range(15):
            t0 = time.perf_counter(); ctx.msm_device(dp, ds, n); ws.append((time.perf_counter() - t0) * 1e3)
            t = ctx.timings()
            for k in ("accumulate", "seg_reduce", "bit_tree"):
                ph[k] = ph.get(k, 0.0) + t[k] / 15
        print(f"2^{logn} chunk_len={L}: median {statistics.median(ws):.3f} ms  min {min(ws):.3f} | accumulate={ph['accumulate']:.3f} fold+rowcol={ph['seg_reduce']:.3f} tree={ph['bit_tree']:.3f}", flush=True)
ctx.set_param("chunk_len", 8)
