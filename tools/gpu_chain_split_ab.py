"""A/B of the two-chain form of one MSM call ("split": high and low half of the windows as two launch chains on two streams) against
the single chain, device-resident inputs, one call after the other; results compared for equality and against the closed form.

    python tools/gpu_chain_split_ab.py [--sizes 16,17,18,19,20,22] [--reps 9]      -> profiles/r04_split_ab.txt
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1


def med(f, reps):
    w = []
    for _ in range(reps):
        t = time.perf_counter(); f(); w.append((time.perf_counter() - t) * 1e3)
    w = sorted(w[1:])
    return w[len(w) // 2], w[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="16,17,18,19,20,22")
    ap.add_argument("--reps", type=int, default=9)
    a = ap.parse_args()
    from curdleproofs_pie_amd import _native as N

    sizes = [int(x) for x in a.sizes.split(",")]
    ctx = N.Context(0)
    nmax = 1 << max(sizes)
    dk, dp, ds, dg = ctx.alloc(32 * nmax), ctx.alloc(96 * nmax), ctx.alloc(32 * nmax), ctx.alloc(96)
    dg.upload(GX.to_bytes(48, "little") + GY.to_bytes(48, "little"))
    ctx.gen_scalars_device(dk, nmax, 1)
    ctx.batch_mul_device(dg, 1, dk, dp, nmax)
    ctx.gen_scalars_device(ds, nmax, 2)
    ctx.set_param("profile", 1)
    print("n        single chain ms (min)     two chains ms (min)      gain     k_accumulate single / two (sum)   host tail on the critical path")
    for lg in sizes:
        n = 1 << lg
        row = {}
        for split in (0, 1):
            ctx.set_param("split", split)
            ctx.set_param("split_min_log2n", 10)          # force it at every size of the sweep
            for _ in range(3):
                out = ctx.msm_device(dp, ds, n)
            m, lo = med(lambda: ctx.msm_device(dp, ds, n), a.reps)
            t = ctx.timings()
            row[split] = (m, lo, out, t["accumulate"], t["host_tail"], ctx.last_counts()["entries"])
        assert N.cg1_eq(row[0][2], row[1][2]) == 1, "split and single-chain results differ at 2^%d" % lg
        assert row[0][5] == row[1][5], "the two forms sort different numbers of bucket entries"
        print("2^%-3d   %8.3f (%7.3f)        %8.3f (%7.3f)      %5.1f %%     %.3f / %.3f                    %.3f / %.3f" % (
            lg, row[0][0], row[0][1], row[1][0], row[1][1], 100.0 * (row[0][0] / row[1][0] - 1.0), row[0][3], row[1][3], row[0][4], row[1][4]))
    ctx.set_param("split", 0)


if __name__ == "__main__":
    main()
