#!/usr/bin/env python3
"""k_rowcol_quad: quads per row / column (16 / 8 / 4; "rowcol_lgq" = 4 / 3 / 2, 0 = the host's cost rule) against the size of the call.

    python tools/gpu_rowcol_ab.py        -> profiles/r04_rowcol_ab.txt
"""
import ctypes
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curdleproofs_pie_amd import _native as N  # noqa: E402

ctx = N.Context(0)
ctx.set_param("profile", 2)
nmax = 1 << 18
dk, dp, ds, dg = ctx.alloc(32 * nmax), ctx.alloc(96 * nmax), ctx.alloc(32 * nmax), ctx.alloc(96)
g = ctypes.create_string_buffer(N.POINT_BYTES); N.cg1_generator(g)
a = ctypes.create_string_buffer(96); N.cg1_to_affine96(a, g.raw)
dg.upload(a.raw)
ctx.gen_scalars_device(dk, nmax, 1)
ctx.batch_mul_device(dg, 1, dk, dp, nmax)
ctx.gen_scalars_device(ds, nmax, 2)
for logn in (12, 13, 14, 15, 16, 17, 18):
    n = 1 << logn
    ref = None
    for lgq in (4, 3, 2, 0):
        ctx.set_param("rowcol_lgq", lgq)
        for _ in range(4):
            out = ctx.msm_device(dp, ds, n)
        ref = ref or out
        assert N.cg1_eq(out, ref) == 1
        ws, seg = [], 0.0
        for _ in range(15):
            t0 = time.perf_counter(); ctx.msm_device(dp, ds, n); ws.append((time.perf_counter() - t0) * 1e3)
            seg += ctx.timings()["seg_reduce"] / 15
        print(f"2^{logn} rowcol_lgq={lgq}: median {statistics.median(ws):.3f} ms  min {min(ws):.3f} | fold + row/column sums = {seg:.3f} ms  c={int(ctx.timings()['window_c'])}", flush=True)
    ctx.set_param("rowcol_lgq", 0)
