#!/usr/bin/env python3
"""A/B of regime A's item sums (1 + hb + lb masked sums per window behind the row / column sums): k_small_tree_quad against
k_small_tree_row (context parameter "tree_row"), wall time and phases of one MSM call, results compared."""
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
    nmax = 1 << 20
    dk, dp, ds, dg = ctx.alloc(32 * nmax), ctx.alloc(96 * nmax), ctx.alloc(32 * nmax), ctx.alloc(96)
    dg.upload(GX.to_bytes(48, "little") + GY.to_bytes(48, "little"))
    ctx.gen_scalars_device(dk, nmax, 1)
    ctx.batch_mul_device(dg, 1, dk, dp, nmax)
    ctx.gen_scalars_device(ds, nmax, 2)
    for logn in (12, 13, 14, 15, 16, 17, 18, 19, 20):
        n = 1 << logn
        res = []
        for tr, rr, fq in ((0, 0, 1), (1, 0, 1), (1, 1, 1), (1, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (1, 1, 0)):
            ctx.set_param("tree_row", tr)
            ctx.set_param("rowcol_row", rr)
            ctx.set_param("fold_quad", fq)
            ctx.set_param("profile", 1)
            w = med(lambda: ctx.msm_device(dp, ds, n, window_c=0))
            ctx.set_param("profile", 2)
            res.append(ctx.msm_device(dp, ds, n, window_c=0))
            tm = ctx.timings()
            print(f"n=2^{logn} tree_row={tr} rowcol_row={rr} fold_quad={fq} c={tm['window_c']}: {w:.3f} ms | seg_reduce={tm['seg_reduce']:.3f} bit_tree={tm['bit_tree']:.3f} host_tail={tm['host_tail']:.3f}" + (f" | same result: {N.cg1_eq(res[0], res[-1]) == 1}" if tr else ""), flush=True)
    for b in (dk, dp, ds, dg):
        b.free()
    ctx.close()


if __name__ == "__main__":
    main()
