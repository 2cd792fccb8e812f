#!/usr/bin/env python3
"""A/B of the endomorphism split (csrc/glv.h, context parameter "glv") over regime-A sizes: wall time and phases of one MSM call with
points + scalars resident in HBM, results compared; then the window plan under the split (window_c sweep) at the mid sizes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N  # noqa: E402

GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1


def med(f, reps=9):
    w = []
    for _ in range(reps):
        t = time.perf_counter(); f(); w.append((time.perf_counter() - t) * 1e3)
    return sorted(w[1:])[len(w[1:]) // 2]


def main():
    import faulthandler
    faulthandler.enable()
    sweep = len(sys.argv) > 1 and sys.argv[1] == "windows"
    sweep_glv = int(os.environ.get("SWEEP_GLV", "1"))
    skip_ab = os.environ.get("SKIP_AB") == "1"
    ctx = N.Context(0)
    nmax = 1 << 22
    dk, dp, ds, dg = ctx.alloc(32 * nmax), ctx.alloc(96 * nmax), ctx.alloc(32 * nmax), ctx.alloc(96)
    dg.upload(GX.to_bytes(48, "little") + GY.to_bytes(48, "little"))
    ctx.gen_scalars_device(dk, nmax, 1)
    ctx.batch_mul_device(dg, 1, dk, dp, nmax)
    ctx.gen_scalars_device(ds, nmax, 2)
    print("## one MSM call, uniform scalars, points in G1 (multiples of the generator): glv 0 / 1", flush=True)
    ab_max = int(os.environ.get("AB_MAX", "22"))
    ab_glv = tuple(int(x) for x in os.environ.get("AB_GLV", "0,1").split(","))
    for logn in (() if skip_ab else tuple(x for x in (12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22) if int(os.environ.get('AB_MIN', '0')) <= x <= ab_max)):
        n = 1 << logn
        row = []
        res = []
        for glv in ab_glv:
            ctx.set_param("glv", 2 if glv else 0)
            ctx.set_param("profile", 1)
            w = med(lambda: ctx.msm_device(dp, ds, n, window_c=0))
            ctx.set_param("profile", 2)
            res.append(ctx.msm_device(dp, ds, n, window_c=0))
            tm = ctx.timings()
            row.append((w, tm))
        same = N.cg1_eq(res[0], res[1]) == 1
        for glv, (w, tm) in enumerate(row):
            print(f"n=2^{logn} glv={glv} c={tm['window_c']}: {w:.3f} ms | " + " ".join(f"{k}={v:.3f}" for k, v in tm.items() if k not in ('window_c', 'host_events')) + ("" if glv == 0 else f" | same result: {same}"), flush=True)
    if sweep:
        print(f"## window plan (glv = {sweep_glv})", flush=True)
        ctx.set_param("glv", 2 if sweep_glv else 0)
        ctx.set_param("profile", 1)
        for logn in tuple(int(x) for x in os.environ.get('SWEEP_N', '13,14,15,16,17,18,19,20').split(',')):
            n = 1 << logn
            out = []
            for c in tuple(int(x) for x in os.environ.get('SWEEP_C', '-11,-12,12,-13,13,-14,14,-15,15,16').split(',')):
                try:
                    out.append((med(lambda: ctx.msm_device(dp, ds, n, window_c=c), reps=7), c))
                except Exception as e:  # noqa: BLE001
                    out.append((float("inf"), c))
            print(f"n=2^{logn}: " + "  ".join(f"c={c}: {w:.3f}" for w, c in out) + f"   best c={min(out)[1]}", flush=True)
    for b in (dk, dp, ds, dg):
        b.free()
    print("buffers freed", flush=True)
    ctx.close()
    print("context closed", flush=True)


if __name__ == "__main__":
    main()
