#!/usr/bin/env python3
"""k_small_tree_quad: lanes per element (block = 4 lanes per element >> "tree_shift") against the size of the call."""
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
for logn in (14, 16, 17, 18, 20):
    n = 1 << logn
    ref = None
    for sh in (1, 2, 3, 4, 0):
        ctx.set_param("tree_shift", sh)
        for _ in range(4):
            out = ctx.msm_device(dp, ds, n)
        ref = ref or out
        assert N.cg1_eq(out, ref) == 1
        ws, bt = [], 0.0
        for _ in range(15):
            t0 = time.perf_counter(); ctx.msm_device(dp, ds, n); ws.append((time.perf_counter() - t0) * 1e3)
            bt += ctx.timings()["bit_tree"] / 15
        print(f"2^{logn} tree_shift={sh}: median {statistics.median(ws):.3f} ms  min {min(ws):.3f} | tree + export = {bt:.3f} ms", flush=True)
    ctx.set_param("tree_shift", 2)
