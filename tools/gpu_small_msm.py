"""One MSM of the protocol's own sizes: wall time per call through the C ABI, k_msm_small (one launch) against the regime-A launch
chain ("small_msm" = 0), with the host-side split of each (enqueue / wait for the GPU / Horner tail).

    python tools/gpu_small_msm.py [--reps 200]          -> profiles/r04_small_msm.txt
"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

R = 52435875175126190479447740508185965837690552500527637822603658699938581184513


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--sizes", default="4,7,32,64,124,128,256,307,512,627,1024,1391,1536,1792,2048")
    ap.add_argument("--sweep-c", default="", help="also time k_msm_small at these window widths (device-resident inputs), e.g. 4,6,7,8,9")
    ap.add_argument("--glv", action="store_true", help="only the A/B of the endomorphism split on the single-launch kernel (device-resident inputs; points are generator multiples)")
    a = ap.parse_args()
    from curdleproofs_pie_amd import _native as N

    ctx = N.Context(0)
    ctx.set_param("profile", 0)
    rng = random.Random(9)
    nmax = 2048
    dk, dg, dp = ctx.alloc(32 * nmax), ctx.alloc(96), ctx.alloc(96 * nmax)
    dk.upload(b"".join(rng.randint(1, R - 1).to_bytes(32, "little") for _ in range(nmax)))
    import ctypes

    g = ctypes.create_string_buffer(144); N.cg1_generator(g)
    g96 = ctypes.create_string_buffer(96); N.cg1_to_affine96(g96, g.raw)
    dg.upload(g96.raw)
    ctx.batch_mul_device(dg, 1, dk, dp, nmax)
    p96 = dp.download()
    if a.glv:
        print("n      glv  c   wall_us   (min)    wait   horner  (k_msm_small, device-resident inputs; same result checked)")
        for n in [int(x) for x in a.sizes.split(",") if int(x) <= 1024]:
            ds = ctx.alloc(32 * n); ds.upload(b"".join(rng.randint(0, R - 1).to_bytes(32, "little") for _ in range(n)))
            ref = None
            for glv in (0, 1, 0, 1):
                ctx.set_param("glv", glv)
                for _ in range(10):
                    out = ctx.msm_device(dp, ds, n)
                ref = ref or out
                assert N.cg1_eq(out, ref) == 1
                wd, wt, hh = [], 0.0, 0.0
                for _ in range(a.reps):
                    t0 = time.perf_counter()
                    ctx.msm_device(dp, ds, n)
                    wd.append((time.perf_counter() - t0) * 1e6)
                    t = ctx.timings()
                    wt += t["host_wait"]; hh += t["host_horner"]
                wd.sort()
                print("%-6d %-4d %-3d %8.1f %7.1f  %6.1f %7.1f" % (n, glv, ctx.timings()["window_c"], wd[len(wd) // 2], wd[0], wt / a.reps * 1e3, hh / a.reps * 1e3), flush=True)
            ds.free()
        return
    print("n      path        c   wall_us   (min)   enqueue  wait   horner   | device-resident inputs: wall_us")
    for n in [int(x) for x in a.sizes.split(",")]:
        s32 = b"".join(rng.randint(0, R - 1).to_bytes(32, "little") for _ in range(n))
        ds = ctx.alloc(32 * n); ds.upload(s32)
        ref = None
        for small in (1, 0):
            ctx.set_param("small_msm", small)
            for _ in range(10):
                out = ctx.msm_host(p96[: 96 * n], s32, n)
            if ref is None:
                ref = out
            assert N.cg1_eq(out, ref) == 1
            ws, parts = [], [0.0, 0.0, 0.0]
            for _ in range(a.reps):
                t0 = time.perf_counter()
                ctx.msm_host(p96[: 96 * n], s32, n)
                ws.append((time.perf_counter() - t0) * 1e6)
                t = ctx.timings()
                parts[0] += t["host_enqueue"]; parts[1] += t["host_wait"]; parts[2] += t["host_horner"]
            wd = []
            for _ in range(a.reps):
                t0 = time.perf_counter()
                ctx.msm_device(dp, ds, n)
                wd.append((time.perf_counter() - t0) * 1e6)
            ws.sort(); wd.sort()
            print("%-6d %-11s %-3d %8.1f %7.1f  %7.1f %6.1f %7.1f   | %8.1f (min %.1f)" % (
                n, "k_msm_small" if small else "regime A", ctx.timings()["window_c"], ws[len(ws) // 2], ws[0],
                parts[0] / a.reps * 1e3, parts[1] / a.reps * 1e3, parts[2] / a.reps * 1e3, wd[len(wd) // 2], wd[0]))
        ctx.set_param("small_msm", 1)
        for c in [int(x) for x in a.sweep_c.split(",") if x]:
            for _ in range(5):
                ctx.msm_device(dp, ds, n, window_c=c)
            wd, wt = [], 0.0
            for _ in range(a.reps):
                t0 = time.perf_counter()
                ctx.msm_device(dp, ds, n, window_c=c)
                wd.append((time.perf_counter() - t0) * 1e6)
                wt += ctx.timings()["host_wait"]
            wd.sort()
            print("%-6d k_msm_small c=%d  wall %.1f us (min %.1f)  wait-for-GPU %.1f us" % (n, c, wd[len(wd) // 2], wd[0], wt / a.reps * 1e3))
        ds.free()
    ctx.set_param("small_msm", 1)


if __name__ == "__main__":
    main()
