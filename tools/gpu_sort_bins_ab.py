#!/usr/bin/env python3
"""Partition sort: sub-bucket width of a k_bin_sort workgroup ("sort_sub_bits") against the size of the call.

    python tools/gpu_sort_bins_ab.py        -> profiles/r04_sort_bins_ab.txt
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
nmax = 1 << 20
dk, dp, ds, dg = ctx.alloc(32 * nmax), ctx.alloc(96 * nmax), ctx.alloc(32 * nmax), ctx.alloc(96)
g = ctypes.create_string_buffer(N.POINT_BYTES); N.cg1_generator(g)
a = ctypes.create_string_buffer(96); N.cg1_to_affine96(a, g.raw)
dg.upload(a.raw)
ctx.gen_scalars_device(dk, nmax, 1)
ctx.batch_mul_device(dg, 1, dk, dp, nmax)
ctx.gen_scalars_device(ds, nmax, 2)
for logn in (13, 14, 15, 16, 17, 18, 19, 20):
    n = 1 << logn
    ref = None
    for sb in (8, 7, 6, 5):
        ctx.set_param("sort_sub_bits", sb)
        for _ in range(4):
            out = ctx.msm_device(dp, ds, n)
        ref = ref or out
        assert N.cg1_eq(out, ref) == 1
        ws, ph = [], {}
        for _ in range(15):
            t0 = time.perf_counter(); ctx.msm_device(dp, ds, n); ws.append((time.perf_counter() - t0) * 1e3)
            t = ctx.timings()
            for k in ("sort_count", "sort_scatter", "chunks"):
                ph[k] = ph.get(k, 0.0) + t[k] / 15
        print(f"2^{logn} sort_sub_bits={sb}: median {statistics.median(ws):.3f} ms  min {min(ws):.3f} | sort_count={ph['sort_count']:.3f} sort_scatter={ph['sort_scatter']:.3f} chunks={ph['chunks']:.3f} c={int(t['window_c'])}", flush=True)
    ctx.set_param("sort_sub_bits", 0)
