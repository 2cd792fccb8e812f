#!/usr/bin/env python3
"""Measurement sweep for DESIGN.md: MSM sizes, structured scalars, emulated per-rank time of the sharded MSM."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N  # noqa: E402

GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def med(f, reps=5):
    w = []
    for _ in range(reps):
        t = time.perf_counter(); f(); w.append((time.perf_counter() - t) * 1e3)
    return sorted(w[1:])[len(w[1:]) // 2]


def main():
    ctx = N.Context(0)
    ctx.set_param("profile", 2)
    nmax = 1 << 23
    dk, dp, ds, dg = ctx.alloc(32 * nmax), ctx.alloc(96 * nmax), ctx.alloc(32 * nmax), ctx.alloc(96)
    dg.upload(GX.to_bytes(48, "little") + GY.to_bytes(48, "little"))
    ctx.gen_scalars_device(dk, nmax, 1)
    t = time.perf_counter(); ctx.batch_mul_device(dg, 1, dk, dp, nmax); print(f"generated 2^23 points in {time.perf_counter()-t:.2f}s", flush=True)
    ctx.gen_scalars_device(ds, nmax, 2)
    print("## size sweep (uniform scalars, whole MSM on one GPU)")
    for logn in (10, 12, 13, 14, 15, 16, 17, 18, 19, 20, 22, 23):
        n = 1 << logn
        ctx.set_param("profile", 1)                                          # wall time without the per-phase marker packets
        w = med(lambda: ctx.msm_device(dp, ds, n, window_c=0), reps=7)      # the library's own plan for the size
        ctx.set_param("profile", 2)
        ctx.msm_device(dp, ds, n, window_c=0)
        tm = ctx.timings()
        print(f"n=2^{logn} c={tm['window_c']}: {w:.3f} ms  {n/w/1e3:.1f} M scalar-mul/s | " + " ".join(f"{k}={v:.3f}" for k, v in tm.items() if k not in ('window_c', 'host_events')), flush=True)
    print("## structured scalars (bucket skew)")
    for agg in (0, 1, 0, 1):
        ctx.set_param("wave_agg", agg)
        w = med(lambda: ctx.msm_device(dp, ds, 1 << 20, window_c=16), reps=9)
        tm = ctx.timings()
        print(f"uniform 2^20 wave_agg={agg}: {w:.3f} ms sort_count={tm['sort_count']:.3f} sort_scatter={tm['sort_scatter']:.3f}", flush=True)
    for agg, logn in ((0, 16), (1, 16), (0, 20), (1, 20)):
        ctx.set_param("wave_agg", agg)
        print("wave_agg =", agg)
        n = 1 << logn
        for name, gen in (("all_equal", lambda i: 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % R), ("sigma_0..n-1", lambda i: i)):
            buf = b"".join(gen(i).to_bytes(32, "little") for i in range(n))
            dss = ctx.alloc(32 * n); dss.upload(buf)
            w = med(lambda: ctx.msm_device(dp, dss, n), reps=3)
            tm = ctx.timings()
            print(f"n=2^{logn} {name}: {w:.2f} ms | " + " ".join(f"{k}={v:.3f}" for k, v in tm.items() if k not in ('host_events',)), flush=True)
            dss.free()
    print("## emulated per-rank time of ONE MSM of N x 2^20 terms (rank 0's share, this GPU)")
    for world in (1, 2, 4, 8):
        n = world << 20
        w = med(lambda: ctx.msm_device(dp, ds, n, window_c=16, shard_rank=0, shard_world=world))
        tm = ctx.timings()
        print(f"windows: world={world} n_total=2^{20 + world.bit_length() - 1}: rank-0 time {w:.3f} ms -> aggregate {n/w/1e3:.1f} M/s | " + " ".join(f"{k}={v:.3f}" for k, v in tm.items() if k not in ('window_c', 'host_events')), flush=True)
    w = med(lambda: ctx.msm_device(dp, ds, 1 << 20, window_c=16))
    print(f"points : any world: per-rank time {w:.3f} ms (each rank runs a full 2^20-term MSM on its shard)", flush=True)


if __name__ == "__main__":
    main()
