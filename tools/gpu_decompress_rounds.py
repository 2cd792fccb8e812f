#!/usr/bin/env python3
"""k_batch_decompress against the number of points: the kernel keeps 3 waves per SIMD (2 with "decompress_waves" = 2), i.e. 3 072
resident waves = 196 608 points per ROUND on 1 024 SIMDs, and every wave lives for the same ~1 ms however many of its 64 lanes hold
a point.  A batch of 1 024 ell = 124 proofs is 599 040 points = 3.047 rounds: the last 0.047 of a round (144 waves) costs most of a
round's time.  This sweep times whole and fractional round counts, with the same-run multiply peak for the fraction.

    python tools/gpu_decompress_rounds.py        -> profiles/r04_decompress_rounds.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from batch_fixture import ShuffleBatch  # noqa: E402
from curdleproofs_pie_amd import _native as N  # noqa: E402
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier  # noqa: E402

MADS_PER_POINT = 148862          # bench.py: 376 squarings x 301 + 85 + 7 products x 392 + the conversions of a decompression


def main():
    fx = ShuffleBatch()
    n = 1100
    inst, proofs, _ = fx.tiled(n)
    ctx = N.Context(0)
    v = ShuffleBatchVerifier(fx.crs, ctx, device_front_end=False)
    L = v.crs.points_per_proof
    points = n * L
    wire = N.PinnedBuffer(ctx, points * 48)
    ctx.check(N.cg1_shuffle_gather_points(v.crs.handle, n, inst, proofs, wire.ptr))
    d_w, d_p, d_s = ctx.alloc(points * 48), ctx.alloc(points * 96), ctx.alloc(points)
    ctx.check(N.cg1_h2d(ctx.handle, d_w.ptr, wire.ptr, points * 48))
    ctx.probe_mad_rate(2, 200)
    peak = max(ctx.probe_mad_rate(2, 100) for _ in range(3))
    print(f"same-run multiply peak: {peak / 1e12:.2f} T/s")
    rnd = 1024 * 3 * 64
    print("points      rounds(3 waves/SIMD)   ms      us per 1024-point block   fraction of the multiply peak")
    for waves in (3, 2):
        ctx.set_param("decompress_waves", waves)
        per_round = 1024 * waves * 64
        for pts in (per_round, 2 * per_round, 3 * per_round, 3 * rnd + 64 * 144, 1024 * L, 3 * per_round + per_round // 2, 4 * per_round // 1 if waves == 2 else 3 * per_round + per_round // 4):
            pts = min(pts, points)
            ctx.check(N.cg1_batch_decompress_device(ctx.handle, d_w.ptr, d_p.ptr, d_s.ptr, pts, 0))
            best = 1e9
            for _ in range(3):
                ctx.timer_begin()
                for _ in range(4):
                    ctx.check(N.cg1_batch_decompress_enqueue(ctx.handle, d_w.ptr, d_p.ptr, d_s.ptr, pts, 0))
                best = min(best, ctx.timer_end() / 4)
            frac = MADS_PER_POINT * pts / (best * 1e-3) / peak
            print(f"{pts:8d}   {pts / per_round:6.3f} ({waves} waves/SIMD)   {best:7.3f}   {best * 1e3 / (pts / 1024):8.3f}                 {frac:.3f}", flush=True)
    ctx.set_param("decompress_waves", 3)


if __name__ == "__main__":
    main()
