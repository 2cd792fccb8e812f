#!/usr/bin/env python3
"""Which window width is fastest per MSM size (single call, latency included)?  Feeds pick_window()."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N  # noqa: E402

GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
ctx = N.Context(0)
ctx.set_param("profile", 2)
nmax = 1 << 20
dk, dp, ds, dg = ctx.alloc(32 * nmax), ctx.alloc(96 * nmax), ctx.alloc(32 * nmax), ctx.alloc(96)
dg.upload(GX.to_bytes(48, "little") + GY.to_bytes(48, "little"))
ctx.gen_scalars_device(dk, nmax, 1); ctx.batch_mul_device(dg, 1, dk, dp, nmax); ctx.gen_scalars_device(ds, nmax, 2)
for n in (4, 64, 627, 1 << 10, 1 << 12, 1 << 13, 1 << 14, 1 << 15, 1 << 16, 1 << 17, 1 << 18, 3 << 17, 1 << 19, 1 << 20):
    res = []
    for c in (0, 4, 8, 13, 16, -8, -9, -10, -11, -12, -13, -14, -15):      # uniform widths; negative = balanced plans (cmax = -c)
        w = []
        for _ in range(6):
            t = time.perf_counter(); ctx.msm_device(dp, ds, n, window_c=c); w.append((time.perf_counter() - t) * 1e3)
        res.append((sorted(w[1:])[2], c if c else "%d" % ctx.timings()["window_c"], c == 0))
    print(f"n={n}: " + "  ".join(f"{'auto->' if a else ''}c{c}:{t:.3f}" for t, c, a in res), flush=True)
ctx.msm_device(dp, ds, 1 << 16)
print("2^16 auto phases:", {k: round(v, 3) for k, v in ctx.timings().items()})
ctx.msm_device(dp, ds, 1 << 18)
print("2^18 auto phases:", {k: round(v, 3) for k, v in ctx.timings().items()})
