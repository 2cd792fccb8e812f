#!/usr/bin/env python3
"""Quick GPU bring-up: parity of the HIP MSM vs the big-int oracle at small n, then timings at large n."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bls12_381 as O  # noqa: E402
from curdleproofs_pie_amd import _native as N  # noqa: E402


def raw96(pt):
    return bytes(96) if pt is None else pt[0].to_bytes(48, "little") + pt[1].to_bytes(48, "little")


def blob_to_affine(blob):
    import ctypes
    out = ctypes.create_string_buffer(96)
    N.cg1_to_affine96(out, blob)
    r = out.raw
    if r == bytes(96):
        return None
    return (int.from_bytes(r[:48], "little"), int.from_bytes(r[48:], "little"))


def main():
    ctx = N.Context(0)
    rng = random.Random(1)
    base_pts = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(64)]
    ok = True
    for n in [1, 2, 3, 7, 64, 200]:
        pts = [base_pts[i % 64] for i in range(n)]
        sc = [rng.randrange(O.R) for _ in range(n)]
        want = O.compute_MSM_fast(pts, sc) if n > 8 else O.compute_MSM(pts, sc)
        for c in ([0, 4, 7, 11, 16] if n <= 64 else [0, 8]):
            t = time.time()
            dp = ctx.alloc(96 * n); ds = ctx.alloc(32 * n)
            dp.upload(b"".join(raw96(p) for p in pts)); ds.upload(b"".join(s.to_bytes(32, "little") for s in sc))
            got = blob_to_affine(ctx.msm_device(dp, ds, n, window_c=c))
            good = got == want
            ok &= good
            print(f"n={n} c={c} {'OK' if good else 'MISMATCH'} {ctx.timings()} {time.time()-t:.3f}s", flush=True)
    # edge cases: zero scalars, identity bases, duplicates, P and -P, all-equal scalars
    pts = [base_pts[0], base_pts[0], O.g1_neg(base_pts[0]), None, base_pts[1], base_pts[1]]
    sc = [5, 5, 10, 7, 0, O.R - 1]
    want = O.compute_MSM(pts, sc)
    dp = ctx.alloc(96 * 6); ds = ctx.alloc(32 * 6)
    dp.upload(b"".join(raw96(p) for p in pts)); ds.upload(b"".join(s.to_bytes(32, "little") for s in sc))
    got = blob_to_affine(ctx.msm_device(dp, ds, 6))
    print("edge:", "OK" if got == want else "MISMATCH", flush=True)
    ok &= got == want
    if not ok:
        print("PARITY FAILED")
        return 1

    # ---- timings with generated points
    for logn in [12, 16, 20]:
        n = 1 << logn
        dsc = ctx.alloc(32 * n)
        ctx.gen_scalars_device(dsc, n, 7)
        dg = ctx.alloc(96); dg.upload(raw96(O.G1_GEN))
        dpts = ctx.alloc(96 * n)
        t = time.time()
        ctx.batch_mul_device(dg, 1, dsc, dpts, n)
        print(f"gen 2^{logn} points: {time.time()-t:.3f}s", flush=True)
        dsc2 = ctx.alloc(32 * n)
        ctx.gen_scalars_device(dsc2, n, 99)
        if logn == 12:
            # spot-check generated points and the MSM against the oracle
            ks = [int.from_bytes(dsc.download(32, 32 * i), "little") for i in range(8)]
            rawp = dpts.download(96 * 8)
            for i in range(8):
                assert rawp[96 * i: 96 * i + 96] == raw96(O.g1_mul(O.G1_GEN, ks[i])), i
            allp = dpts.download()
            alls = dsc2.download()
            P = [(int.from_bytes(allp[96 * i: 96 * i + 48], "little"), int.from_bytes(allp[96 * i + 48: 96 * i + 96], "little")) for i in range(n)]
            S = [int.from_bytes(alls[32 * i: 32 * i + 32], "little") for i in range(n)]
            t = time.time()
            want = O.compute_MSM_fast(P, S, c=8)
            print(f"oracle 2^12 bucket MSM {time.time()-t:.1f}s", flush=True)
            got = blob_to_affine(ctx.msm_device(dpts, dsc2, n))
            print("2^12 parity:", "OK" if got == want else "MISMATCH", flush=True)
        for c in ([0] if logn < 16 else [0, 13, 14, 15, 16]):
            for rep in range(3):
                t = time.time()
                ctx.msm_device(dpts, dsc2, n, window_c=c)
                wall = time.time() - t
            tm = ctx.timings()
            print(f"MSM 2^{logn} c={tm['window_c']}: wall {wall*1e3:.2f} ms -> {n/wall/1e6:.2f} M scalar-mul/s | " +
                  " ".join(f"{k}={v:.3f}" for k, v in tm.items() if k != "window_c"), flush=True)
        if logn == 16:
            for lanes, iters in [(65536, 32), (131072, 32), (262144, 32)]:
                ms = ctx.probe_madd(dpts, 1024, lanes, iters)
                print(f"probe madd lanes={lanes} iters={iters}: {ms:.3f} ms -> {lanes*iters/ms/1e6:.2f} G madd/s", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
