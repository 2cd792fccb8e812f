#!/usr/bin/env python3
"""Timing breakdown of one MSM size on the GPU (phases, host overheads).  usage: gpu_timing.py [logn] [c] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N  # noqa: E402

GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    cs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    ctx = N.Context(0)
    n = 1 << logn
    dk, dp, ds, dg = ctx.alloc(32 * n), ctx.alloc(96 * n), ctx.alloc(32 * n), ctx.alloc(96)
    dg.upload(GX.to_bytes(48, "little") + GY.to_bytes(48, "little"))
    ctx.gen_scalars_device(dk, n, 1)
    ctx.batch_mul_device(dg, 1, dk, dp, n)
    ctx.gen_scalars_device(ds, n, 2)
    variants = [("default", {})]
    if os.environ.get("AB"):      # e.g. AB=reduce_2d=0,reduce_2d=1  -> interleaved A/B in one process
        variants = [(kv, {kv.split("=")[0]: int(kv.split("=")[1])}) for kv in os.environ["AB"].split(",")] * 2
    for name, params in variants:
        for k, v in params.items():
            ctx.set_param(k, v)
        for c in cs:
            walls = []
            acc = {}
            for r in range(reps):
                t = time.perf_counter()
                ctx.msm_device(dp, ds, n, window_c=c)
                walls.append((time.perf_counter() - t) * 1e3)
                if r >= 2:
                    for k, v in ctx.timings().items():
                        acc[k] = acc.get(k, 0) + v / (reps - 2)
            w = sorted(walls[2:])
            print(f"[{name}] 2^{logn} c={c}: wall min {w[0]:.3f} med {w[len(w)//2]:.3f} ms -> {n/w[len(w)//2]/1e3:.1f} M/s | " +
                  " ".join(f"{k}={v:.3f}" for k, v in acc.items() if k != "window_c"), flush=True)


if __name__ == "__main__":
    main()
