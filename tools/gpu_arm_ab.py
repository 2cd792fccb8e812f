#!/usr/bin/env python3
"""Host tail with the Horner's helper threads armed (spinning for their part while the GPU result is polled) against woken by futex."""
import ctypes, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curdleproofs_pie_amd import _native as N  # noqa: E402
ctx = N.Context(0)
nmax = 1 << 18
dk, dp, ds, dg = ctx.alloc(32 * nmax), ctx.alloc(96 * nmax), ctx.alloc(32 * nmax), ctx.alloc(96)
g = ctypes.create_string_buffer(N.POINT_BYTES); N.cg1_generator(g)
a = ctypes.create_string_buffer(96); N.cg1_to_affine96(a, g.raw)
dg.upload(a.raw)
ctx.gen_scalars_device(dk, nmax, 1)
ctx.batch_mul_device(dg, 1, dk, dp, nmax)
ctx.gen_scalars_device(ds, nmax, 2)
for n in (124, 627, 2048, 1 << 14, 1 << 16, 1 << 18):
    ref = None
    for arm in (0, 1, 0, 1):
        ctx.set_param("arm_helpers", arm)
        for _ in range(6):
            out = ctx.msm_device(dp, ds, n)
        ref = ref or out
        assert N.cg1_eq(out, ref) == 1
        ws, hz = [], 0.0
        for _ in range(40):
            t0 = time.perf_counter(); ctx.msm_device(dp, ds, n); ws.append((time.perf_counter() - t0) * 1e3)
            hz += ctx.timings()["host_horner"] / 40
        print(f"n={n:7d} arm_helpers={arm}: median {statistics.median(ws):.3f} ms  min {min(ws):.3f} | host Horner {hz:.3f} ms", flush=True)
ctx.set_param("arm_helpers", 1)
