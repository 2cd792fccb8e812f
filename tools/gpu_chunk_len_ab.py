#!/usr/bin/env python3
"""Minimum chunk length ("chunk_len") against the size of a regime-A call: wall time of one MSM, inputs resident."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N  # noqa: E402

GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1


def med(f, reps=11):
    w = []
    for _ in range(reps):
        t = time.perf_counter(); f(); w.append((time.perf_counter() - t) * 1e3)
    return sorted(w[1:])[len(w[1:]) // 2]


def main():
    ctx = N.Context(0)
    nmax = 1 << 18
    dk, dp, ds, dg = ctx.alloc(32 * nmax), ctx.alloc(96 * nmax), ctx.alloc(32 * nmax), ctx.alloc(96)
    dg.upload(GX.to_bytes(48, "little") + GY.to_bytes(48, "little"))
    ctx.gen_scalars_device(dk, nmax, 1)
    ctx.batch_mul_device(dg, 1, dk, dp, nmax)
    ctx.gen_scalars_device(ds, nmax, 2)
    ctx.set_param("chunk_rule", int(os.environ.get("CHUNK_RULE", "1")))
    for logn in (13, 14, 15, 16, 17, 18):
        n = 1 << logn
        out = []
        for L in (4, 6, 8, 10, 12, 16, 20, 24, 32, 48):
            ctx.set_param("chunk_len", L)
            ctx.set_param("profile", 1)
            w = med(lambda: ctx.msm_device(dp, ds, n, window_c=0))
            out.append((w, L))
        print(f"n=2^{logn}: " + "  ".join(f"L={L}: {w:.3f}" for w, L in out) + f"   best L={min(out)[1]}", flush=True)
    for b in (dk, dp, ds, dg):
        b.free()
    ctx.close()


if __name__ == "__main__":
    main()
