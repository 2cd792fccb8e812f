"""A/B of the three ways to run a chain of DEPENDENT EC additions on the GPU (csrc/fp_row.h, k_probe_add_chain): one lane per addition
(the formulas of k_accumulate), one DPP quad per addition (g1_quad.h: what the latency-bound kernels use), one LIMB per lane with the
four products of a stage on the four rows of a wave (round 5).  Reports the time per addition for a lone wave and for a full chip, and
checks every mode's result against the host library's chain of the same additions.

    python tools/gpu_rowlane_ab.py [--iters 400]            -> profiles/r05_rowlane_ab.txt
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def host_chain(N, p0, p1, iters):
    """acc = P0; acc += (P1, P0 alternating) `iters` times, on the host library's operators"""
    b0, b1 = ctypes.create_string_buffer(144), ctypes.create_string_buffer(144)
    assert N.cg1_from_affine96(b0, p0, 1) == 0 and N.cg1_from_affine96(b1, p1, 1) == 0
    acc = b0.raw
    out = ctypes.create_string_buffer(144)
    for i in range(iters):
        N.cg1_add(out, acc, b0.raw if (i & 1) else b1.raw)
        acc = out.raw
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from curdleproofs_pie_amd import _native as N

    N.tune_runtime()
    ctx = N.Context(0)
    g = ctypes.create_string_buffer(144)
    N.cg1_generator(g)
    pts = []
    for k in (0x1234567, 0x7654321):
        b = ctypes.create_string_buffer(144)
        N.cg1_mul(b, g.raw, k.to_bytes(32, "little"))
        o = ctypes.create_string_buffer(96)
        N.cg1_to_affine96(o, b.raw)
        pts.append(o.raw)
    two = pts[0] + pts[1]
    want = host_chain(N, pts[0], pts[1], a.iters)
    names = {0: "one lane per addition (xyzz_add)", 1: "one DPP quad per addition (quad_add)", 2: "one limb per lane, 4 rows = 4 products (row_add)"}
    res = {"iters": a.iters, "what": "device time of ONE launch / iters = time per dependent addition of a wave's chain", "modes": {}}
    for mode in (0, 1, 2):
        rec = {"name": names[mode]}
        for waves, label in ((1, "lone_wave"), (256, "one_wave_per_cu"), (1024, "one_wave_per_simd"), (2048, "two_waves_per_simd"), (8192, "eight_per_simd")):
            out = ctypes.create_string_buffer(144)
            ms = ctypes.c_float(0)
            ctx.check(N.cg1_probe_add_chain(ctx.handle, mode, two, waves, a.iters, a.reps, out, ctypes.byref(ms)))
            assert N.cg1_eq(out.raw, want) == 1, "mode %d differs from the host chain" % mode
            adds_per_wave = {0: 64, 1: 16, 2: 1}[mode]          # independent additions a wave COULD carry in this mode
            rec[label] = {"ms": round(ms.value, 4), "us_per_dependent_add": round(ms.value * 1e3 / a.iters, 3),
                          "chip_adds_per_us_if_all_lanes_differ": round(waves * adds_per_wave * a.iters / (ms.value * 1e3), 1)}
        res["modes"][str(mode)] = rec
    # exceptional cases through the row path: P + P, P - P, identity operands are decided by the one-lane formulas
    lone = {m: res["modes"][str(m)]["lone_wave"]["us_per_dependent_add"] for m in (0, 1, 2)}
    res["lone_wave_speedup_row_vs_quad"] = round(lone[1] / lone[2], 2)
    res["lone_wave_speedup_row_vs_lane"] = round(lone[0] / lone[2], 2)
    res["applications"] = applications(N, ctx)
    print(json.dumps(res, indent=1))


def applications(N, ctx):
    """The two places one limb per lane is switched into, A/B on this box: regime B's per-MSM Horner (1 024 independent 627-term MSMs:
    BASELINE config 3's shape) and a deferred map batch of 248 `R * k` results (the instance of one N = 128 shuffle proof,
    curdleproofs.py:310-311) through cg1_lincomb_batch."""
    import random
    import time

    out = {}
    rng = random.Random(7)
    g = ctypes.create_string_buffer(144)
    N.cg1_generator(g)
    n_base = 627
    recs = []
    for i in range(n_base):
        b = ctypes.create_string_buffer(144)
        N.cg1_mul(b, g.raw, rng.randrange(1, 2 ** 250).to_bytes(32, "little"))
        o = ctypes.create_string_buffer(96)
        N.cg1_to_affine96(o, b.raw)
        recs.append(o.raw)
    m_msm = 1024
    pts = b"".join(recs) * m_msm
    sc = b"".join(rng.randrange(1, 2 ** 254).to_bytes(32, "little") for _ in range(n_base * m_msm))
    d_p, d_s = ctx.alloc(len(pts)), ctx.alloc(len(sc))
    d_p.upload(pts); d_s.upload(sc)
    offsets = [n_base * j for j in range(m_msm + 1)]
    rb = {}
    first = {}
    for flag in (0, 1, 0, 1):
        ctx.set_param("horner_row", flag)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            blobs = ctx.msm_batched_device(d_p, d_s, offsets)
            best = min(best, (time.perf_counter() - t0) * 1e3)
        rb[flag] = min(rb.get(flag, 1e9), best)
        first[flag] = blobs[0]
    ctx.set_param("horner_row", 1)
    assert N.cg1_eq(first[0], first[1]) == 1
    out["regime_b_1024_x_627_ms"] = {"horner_one_quad_per_msm": round(rb[0], 3), "horner_one_wave_per_msm_rows": round(rb[1], 3)}
    d_p.free(); d_s.free()
    # 248 results s * B
    n_out = 248
    raw = b"".join(recs[:n_out])
    offs = (ctypes.c_uint32 * (n_out + 1))(*range(n_out + 1))
    tba = (ctypes.c_uint32 * n_out)(*range(n_out))
    k = rng.randrange(1, 2 ** 254).to_bytes(32, "little")
    scb = k * n_out
    mm = {}
    ref = None
    for label, path, row in (("host_pool", 1, 1), ("k_batch_mul_row", 0, 1), ("host_pool", 1, 1), ("k_batch_mul_row", 0, 1)):
        ctx.set_param("batch_mul_row", row)
        best = 1e9
        for _ in range(5):
            ob = ctypes.create_string_buffer(144 * n_out)
            used = ctypes.c_int(0)
            t0 = time.perf_counter()
            ctx.check(N.cg1_lincomb_batch(ctx.handle, raw, n_out, offs, n_out, tba, scb, path, ob, None, None, ctypes.byref(used)))
            best = min(best, (time.perf_counter() - t0) * 1e3)
        mm[label] = min(mm.get(label, 1e9), best)
        if ref is None:
            ref = ob.raw
        assert ob.raw == ref
    out["map_of_248_results_ms"] = {k2: round(v, 3) for k2, v in mm.items()}
    out["host_pool_threads"] = int(N.cg1_shuffle_default_threads())
    # one small MSM (k_msm_small) with its items / combine / export on quads against one wave per item with one limb per lane
    sm = {}
    for n in (4, 124, 256, 627, 1024, 2048):
        pts = b"".join(recs[i % n_base] for i in range(n))
        scs = b"".join(rng.randrange(1, 2 ** 254).to_bytes(32, "little") for _ in range(n))
        d_p, d_s = ctx.alloc(len(pts)), ctx.alloc(len(scs))
        d_p.upload(pts); d_s.upload(scs)
        row = {}
        ref = None
        for flag in (0, 1, 0, 1):
            ctx.set_param("small_row_tail", flag)
            best = 1e9
            for _ in range(30):
                t0 = time.perf_counter()
                blob = ctx.msm_device(d_p, d_s, n)
                best = min(best, (time.perf_counter() - t0) * 1e3)
            row[flag] = min(row.get(flag, 1e9), best)
            if ref is None:
                ref = blob
            assert N.cg1_eq(ref, blob) == 1
        sm[str(n)] = {"quads_ms": round(row[0], 4), "rows_ms": round(row[1], 4)}
        d_p.free(); d_s.free()
    ctx.set_param("small_row_tail", 1)
    out["one_small_msm_device_resident_ms"] = sm
    return out


if __name__ == "__main__":
    main()
