#!/usr/bin/env python3
"""Timing of regime B: M independent n-term MSMs in one launch chain.  usage: gpu_timing_batched.py [M] [n] [c,...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N  # noqa: E402

GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 627
    cs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
    ctx = N.Context(0)
    tot = M * n
    dk, dp, ds, dg = ctx.alloc(32 * tot), ctx.alloc(96 * tot), ctx.alloc(32 * tot), ctx.alloc(96)
    dg.upload(GX.to_bytes(48, "little") + GY.to_bytes(48, "little"))
    ctx.gen_scalars_device(dk, tot, 1)
    ctx.batch_mul_device(dg, 1, dk, dp, tot)
    ctx.gen_scalars_device(ds, tot, 2)
    offs = [n * j for j in range(M + 1)]
    glvs = [int(x) for x in os.environ.get("GLV", "0").split(",")]
    splits = [int(x) for x in os.environ.get("BSPLIT", "1").split(",")]
    for split, glv, c in [(sp, g, c) for sp in splits for g in glvs for c in cs]:
        ctx.set_param("glv", glv)
        ctx.set_param("batched_split", split)
        walls = []
        for r in range(5):
            t = time.perf_counter()
            ctx.msm_batched_device(dp, ds, offs, window_c=c)
            walls.append((time.perf_counter() - t) * 1e3)
        tm = ctx.timings()
        w = sorted(walls[1:])[len(walls[1:]) // 2]
        print(f"batched {M} x {n} two_chains={split} glv={glv} c={tm['window_c']}: wall {w:.2f} ms -> {M/w*1e3:.0f} MSM/s, {tot/w/1e3:.1f} M scalar-mul/s | " +
              " ".join(f"{k}={v:.3f}" for k, v in tm.items() if k != "window_c"), flush=True)
    # serial single-MSM calls for comparison
    ctx.set_param("glv", 0)
    t = time.perf_counter()
    for j in range(32):
        ctx.msm_device(dp.ptr + 96 * n * j, ds.ptr + 32 * n * j, n)
    dt = (time.perf_counter() - t) / 32 * 1e3
    print(f"single-MSM path, n={n}: {dt:.3f} ms per MSM (c={ctx.timings()['window_c']}) -> {1e3/dt:.0f} MSM/s", flush=True)


if __name__ == "__main__":
    main()
