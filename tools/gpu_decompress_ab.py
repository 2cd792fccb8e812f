#!/usr/bin/env python3
"""Same-box A/B of k_batch_decompress<false> compiled for 2 and for 3 waves per SIMD ("decompress_waves"): one launch over the 599 040
wire points of a 1024-proof batch (hipEvents on the stream), and the verify stream with either setting."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from batch_fixture import ShuffleBatch
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

fx = ShuffleBatch()
n = 1024
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
ref = None
for rnd in range(3):
    for waves in (2, 3):
        ctx.set_param("decompress_waves", waves)
        ctx.check(N.cg1_batch_decompress_device(ctx.handle, d_w.ptr, d_p.ptr, d_s.ptr, points, 0))
        ctx.timer_begin()
        for _ in range(5):
            ctx.check(N.cg1_batch_decompress_enqueue(ctx.handle, d_w.ptr, d_p.ptr, d_s.ptr, points, 0))
        ms = ctx.timer_end() / 5
        out = d_p.download()
        ref = ref or out
        assert out == ref and not any(d_s.download(points))
        print(f"k_batch_decompress<false>, {waves} waves per SIMD: {ms:.3f} ms per {points} points", flush=True)
peak = max(ctx.probe_mad_rate(2, 100) for _ in range(3))
print(f"same-run multiply peak: {peak / 1e12:.2f} T/s", flush=True)
for rnd in range(2):
    for waves in (2, 3):
        for fe in (False, True):
            vv = ShuffleBatchVerifier(fx.crs, ctx, device_front_end=fe)
            vv.ctx.set_param("decompress_waves", waves)
            list(vv.verify_stream([(inst, proofs, n)] * 6))
            K = 30
            t0 = time.perf_counter()
            for st in vv.verify_stream(((inst, proofs, n) for _ in range(K))):
                assert not any(st)
            dt = time.perf_counter() - t0
            print(f"verify stream, {waves} waves per SIMD, front-end on the {'device' if fe else 'host'}: {1e3 * dt / K:.2f} ms per batch = {n * K / dt:.0f} proofs/s", flush=True)
            vv.close()
ctx.set_param("decompress_waves", 3)
