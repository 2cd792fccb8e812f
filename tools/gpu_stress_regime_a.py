#!/usr/bin/env python3
"""Stress of the regime-A and single-launch paths: thousands of back-to-back MSM calls of random sizes and window plans over random
slices of one point table, every result compared with the closed form (sum k_i s_i) G computed on the host (points are k_i G).
Exercises the zero-copy export (ticket + flag word written by the last block of k_small_tree_row) call after call.

    python tools/gpu_stress_regime_a.py [seconds] [seed]
"""
import ctypes
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N  # noqa: E402

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    ctx = N.Context(0)
    nmax = 1 << 17
    ks = [rng.randrange(1, 1 << 64) for _ in range(nmax)]
    ss = [rng.randrange(0, R) for _ in range(nmax)]
    for i in range(0, nmax, 997):
        ss[i] = rng.choice([0, 1, R - 1, (R + 1) // 2, rng.randrange(0, 1 << 128)])
    g = ctypes.create_string_buffer(144); N.cg1_generator(g)
    g96 = ctypes.create_string_buffer(96); N.cg1_to_affine96(g96, g.raw)
    dk, dp, ds, dg = ctx.alloc(32 * nmax), ctx.alloc(96 * nmax), ctx.alloc(32 * nmax), ctx.alloc(96)
    dg.upload(g96.raw)
    dk.upload(b"".join(k.to_bytes(32, "little") for k in ks))
    ds.upload(b"".join(s.to_bytes(32, "little") for s in ss))
    ctx.batch_mul_device(dg, 1, dk, dp, nmax)
    # prefix sums of k_i s_i: the closed form of any slice in O(1)
    pre = [0]
    for k, s in zip(ks, ss):
        pre.append((pre[-1] + k * s) % R)
    t_end = time.time() + budget
    calls = 0
    by_path = {}
    while time.time() < t_end:
        logn = rng.choice([0, 2, 5, 8, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 16, 17])
        n = max(1, min(nmax, rng.randrange(1 << logn, (2 << logn))))
        lo = rng.randrange(0, nmax - n + 1)
        c = rng.choice([0, 0, 0, 0, 8, 11, -12, -13, -14, 16]) if n > 2048 else 0
        for name, val in (("tree_row", rng.choice([1, 1, 1, 0])), ("rowcol_row", rng.choice([1, 1, 0])), ("glv", rng.choice([0, 0, 1, 2]))):
            ctx.set_param(name, val)
        out = ctx.msm_device(dp.ptr + 96 * lo, ds.ptr + 32 * lo, n, window_c=c)
        want = ctypes.create_string_buffer(144)
        N.cg1_mul(want, g.raw, ((pre[lo + n] - pre[lo]) % R).to_bytes(32, "little"))
        if N.cg1_eq(out, want.raw) != 1:
            print(f"MISMATCH after {calls} calls: n={n} lo={lo} c={c}", flush=True)
            sys.exit(1)
        calls += 1
        key = "small" if n <= 2048 else "regime A"
        by_path[key] = by_path.get(key, 0) + 1
        if calls % 2000 == 0:
            print(f"{calls} calls ok ({by_path})", flush=True)
    print(f"stress ok: {calls} calls in {budget:.0f} s, seed {seed}: {by_path}", flush=True)
    for b in (dk, dp, ds, dg):
        b.free()
    ctx.close()


if __name__ == "__main__":
    main()
