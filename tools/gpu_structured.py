#!/usr/bin/env python3
"""Structured scalars at 2^20 (bucket skew): all-equal and sigma = 0..n-1.  Run under rocprofv3 for the kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ctx = N.Context(0)
n = 1 << 20
g = __import__("ctypes").create_string_buffer(144); N.cg1_generator(g)
a = __import__("ctypes").create_string_buffer(96); N.cg1_to_affine96(a, g.raw)
dg, dk, dp = ctx.alloc(96), ctx.alloc(32 * n), ctx.alloc(96 * n)
dg.upload(a.raw); ctx.gen_scalars_device(dk, n, 5); ctx.batch_mul_device(dg, 1, dk, dp, n)
for name, gen in (("all_equal", lambda i: 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % R), ("sigma", lambda i: i),
                  ("two_values", lambda i: (3, R - 5)[i & 1]), ("low_32_bits", lambda i: (i * 2654435761) & 0xFFFFFFFF)):
    ds = ctx.alloc(32 * n); ds.upload(b"".join(gen(i).to_bytes(32, "little") for i in range(n)))
    ts = []
    for _ in range(4):
        t0 = time.perf_counter(); ctx.msm_device(dp, ds, n); ts.append(time.perf_counter() - t0)
    tm = ctx.timings()
    print(f"{name}: {1e3*min(ts):.2f} ms | " + " ".join(f"{k}={v:.3f}" for k, v in tm.items() if k in ("sort_count", "sort_scatter", "chunks", "accumulate", "seg_reduce", "bit_tree")), flush=True)
    ds.free()
