"""One limb per lane (csrc/fp_row.h): a chain of dependent EC additions run by the three device formulations -- one lane, a DPP quad,
one limb per lane on the four rows of a wave -- ends in the point the host library's chain ends in, which is the oracle's
(k0 + a k1 + b k0) G for the multiples of the generator the chain adds up; the exceptional additions (P + P, P - P, the identity)
go through the same entry point."""
import ctypes

import pytest

from oracle import bls12_381 as O

pytestmark = pytest.mark.gpu


def _affine96(pt):
    return bytes(96) if pt is None else pt[0].to_bytes(48, "little") + pt[1].to_bytes(48, "little")


@pytest.mark.parametrize("iters", [0, 1, 2, 7, 64])
def test_three_formulations_agree_with_the_oracle(native_lib, iters):
    N = native_lib
    ctx = N.default_context()
    k0, k1 = 0x1234567, 0x7654321
    p0, p1 = O.g1_mul(O.G1_GEN, k0), O.g1_mul(O.G1_GEN, k1)
    # acc = P0; acc += (P1, P0, P1, ...)
    tot = k0 + sum(k0 if (i & 1) else k1 for i in range(iters))
    want = O.g1_compress(O.g1_mul(O.G1_GEN, tot % O.R))
    for mode in (0, 1, 2):
        out = ctypes.create_string_buffer(144)
        ms = ctypes.c_float(0)
        ctx.check(N.cg1_probe_add_chain(ctx.handle, mode, _affine96(p0) + _affine96(p1), 3, iters, 1, out, ctypes.byref(ms)))
        c = ctypes.create_string_buffer(48)
        N.cg1_compress(c, out.raw)
        assert c.raw == want, (mode, iters)


def test_exceptional_additions_on_rows(native_lib):
    """P0 == P1 makes the first addition a doubling; P1 == -P0 makes the chain pass through the identity every other step."""
    N = native_lib
    ctx = N.default_context()
    p0 = O.g1_mul(O.G1_GEN, 99)
    for p1, coef in ((p0, 1), (O.g1_neg(p0), -1)):
        for iters in (1, 2, 3, 6):
            tot = 99 + sum(99 if (i & 1) else 99 * coef for i in range(iters))
            want = O.g1_compress(O.g1_mul(O.G1_GEN, tot % O.R) if tot % O.R else None)
            for mode in (0, 1, 2):
                out = ctypes.create_string_buffer(144)
                ms = ctypes.c_float(0)
                ctx.check(N.cg1_probe_add_chain(ctx.handle, mode, _affine96(p0) + _affine96(p1), 2, iters, 1, out, ctypes.byref(ms)))
                c = ctypes.create_string_buffer(48)
                N.cg1_compress(c, out.raw)
                assert c.raw == want, (mode, iters, coef)


def test_regime_b_horner_one_wave_per_msm(native_lib):
    """cg1_msm_batched with more MSMs than finish on the host (> 24): the per-MSM Horner on the device, one wave per MSM with one limb
    per lane ("horner_row" = 1, the default) and one quad per MSM (= 0): the same points, and the oracle's on a sample."""
    import random

    N = native_lib
    ctx = N.default_context()
    rng = random.Random(41)
    m_msm = 40
    offsets, pts, sc, ks = [0], [], [], []
    base_pts = [(k, O.g1_mul(O.G1_GEN, k)) for k in (rng.randrange(1, O.R) for _ in range(16))]
    for j in range(m_msm):
        n = rng.choice([1, 3, 17, 64])
        for _ in range(n):
            k, p = base_pts[rng.randrange(16)]
            ks.append(k)
            pts.append(_affine96(p))
            sc.append(rng.randrange(O.R))
        offsets.append(len(pts))
    raw_p, raw_s = b"".join(pts), b"".join(s.to_bytes(32, "little") for s in sc)
    got = {}
    for flag in (1, 0):
        ctx.set_param("horner_row", flag)
        try:
            got[flag] = ctx.msm_batched_host(raw_p, raw_s, offsets)
        finally:
            ctx.set_param("horner_row", 1)
    for j in range(m_msm):
        assert N.cg1_eq(got[0][j], got[1][j]) == 1, j
    for j in rng.sample(range(m_msm), 6):
        tot = sum(ks[t] * sc[t] for t in range(offsets[j], offsets[j + 1])) % O.R
        c = ctypes.create_string_buffer(48)
        N.cg1_compress(c, got[1][j])
        assert c.raw == O.g1_compress(O.g1_mul(O.G1_GEN, tot) if tot else None)


def test_map_and_fold_batches_on_one_wave_per_result(native_lib):
    """A deferred batch of `s * B` / `A + s * B` results (the callers' map and fold loops) through cg1_lincomb_batch: k_batch_mul_row
    (one wave per result) gives what the host's pool gives, byte for byte; points outside G1, the identity, zero and unit scalars
    included."""
    import random

    N = native_lib
    ctx = N.default_context()
    rng = random.Random(42)
    T3 = (0, 2)
    bases = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(30)] + [None, O.g1_add(O.g1_mul(O.G1_GEN, 5), T3), T3]
    raw = b"".join(_affine96(p) for p in bases)
    offsets, tb, sc = [0], [], []
    for j in range(300):
        kind = j % 5
        if kind in (0, 1, 2):                                    # s * B
            tb.append(rng.randrange(len(bases)) | (0x80000000 if rng.random() < 0.3 else 0)); sc.append(rng.choice([rng.randrange(2, O.R), O.R - 1, 2, 3]))
        elif kind == 3:                                          # A + s * B
            tb.append(rng.randrange(len(bases))); sc.append(1)
            tb.append(rng.randrange(len(bases)) | (0x80000000 if rng.random() < 0.3 else 0)); sc.append(rng.randrange(2, O.R))
        else:                                                    # s * B + A with A = +-B (the addition is a doubling / cancels)
            b = rng.randrange(30)
            tb.append(b); sc.append(2)
            tb.append(b | (0x80000000 if rng.random() < 0.5 else 0)); sc.append(1)
        offsets.append(len(tb))
    n_out = len(offsets) - 1
    offs = (ctypes.c_uint32 * (n_out + 1))(*offsets)
    tba = (ctypes.c_uint32 * len(tb))(*tb)
    scb = b"".join(s.to_bytes(32, "little") for s in sc)
    outs = {}
    for label, path, row in (("pool", 1, 1), ("row_kernel", 0, 1), ("without_row_kernel", 0, 0)):
        ctx.set_param("batch_mul_row", row)
        try:
            ob, oa, ok = (ctypes.create_string_buffer(144 * n_out), ctypes.create_string_buffer(96 * n_out), ctypes.create_string_buffer(48 * n_out))
            used = ctypes.c_int(0)
            ctx.check(N.cg1_lincomb_batch(ctx.handle, raw, len(bases), offs, n_out, tba, scb, path, ob, oa, ok, ctypes.byref(used)))
            outs[label] = (ob.raw, oa.raw, ok.raw, used.value)
        finally:
            ctx.set_param("batch_mul_row", 1)
    assert outs["row_kernel"][3] == 2 and outs["pool"][3] == 1
    assert outs["pool"][:3] == outs["row_kernel"][:3] == outs["without_row_kernel"][:3]


@pytest.mark.parametrize("n,c", [(3000, 0), (1 << 13, 8), (1 << 14, -12), (1 << 14, 5), (1 << 15, -13), (1 << 16, 16), (70000, -15), (1 << 17, 11)])
def test_item_sums_on_rows_equal_the_quads(n, c):
    """Regime A's 1 + hb + lb item sums per window: k_small_tree_row ("tree_row" = 1, the default) against k_small_tree_quad, and both
    against the closed form (sum k_i s_i) G for points k_i G."""
    import numpy as np
    from curdleproofs_pie_amd import _native as N
    from oracle import bls12_381 as O

    ctx = N.Context(0)
    rng = np.random.default_rng(n + 7 * abs(c))
    gen96 = O.G1_GEN[0].to_bytes(48, "little") + O.G1_GEN[1].to_bytes(48, "little")
    m = 256
    ks = [int(x) for x in rng.integers(1, 1 << 62, m)]
    base = ctx.batch_mul_add_host(gen96, 1, b"".join(k.to_bytes(32, "little") for k in ks), m, None, m)
    idx = rng.integers(0, m, n)
    pts = np.frombuffer(base, dtype=np.uint8).reshape(m, 96)[idx].copy()
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x3F
    d_p = ctx.alloc(n * 96); d_p.upload(pts.tobytes())
    d_s = ctx.alloc(n * 32); d_s.upload(sc.tobytes())
    ctx.set_param("small_msm", 0)
    ctx.set_param("tree_row", 0)
    want = ctx.msm_device(d_p, d_s, n, window_c=c)
    ctx.set_param("tree_row", 1)
    ctx.set_param("rowcol_row", 0)
    mid = ctx.msm_device(d_p, d_s, n, window_c=c)
    ctx.set_param("rowcol_row", 1)                          # (small bucket counts: the row / column sums' cross-quad levels on rows too)
    got = ctx.msm_device(d_p, d_s, n, window_c=c)
    assert N.cg1_eq(got, want) == 1 and N.cg1_eq(mid, want) == 1
    tot = sum(ks[i] * int.from_bytes(s.tobytes(), "little") for i, s in zip(idx, sc)) % O.R
    ref = O.g1_mul(O.G1_GEN, tot)
    out = ctypes.create_string_buffer(96)
    N.cg1_to_affine96(out, got)
    assert out.raw == ref[0].to_bytes(48, "little") + ref[1].to_bytes(48, "little")
    ctx.close()


@pytest.mark.parametrize("n", [2500, 5000, 40000, 1 << 17])
def test_row_form_tails_meet_doublings_and_cancellations(n):
    """A handful of distinct points and their negatives under a handful of distinct scalars: whole buckets, rows and columns hold EQUAL or
    OPPOSITE sums, so the row-form additions of k_rowcol_quad_row / k_small_tree_row run into P + P and P - P all the time (their
    out-of-line exceptional path).  Rows against quads against the closed form."""
    import numpy as np
    from curdleproofs_pie_amd import _native as N
    from oracle import bls12_381 as O

    ctx = N.Context(0)
    rng = np.random.default_rng(n)
    ks = [5, 7, O.R - 5, O.R - 7, 5, 11, O.R - 11, 5]
    gen96 = O.G1_GEN[0].to_bytes(48, "little") + O.G1_GEN[1].to_bytes(48, "little")
    base = ctx.batch_mul_add_host(gen96, 1, b"".join(k.to_bytes(32, "little") for k in ks), len(ks), None, len(ks))
    idx = np.arange(n) % len(ks)
    pts = np.frombuffer(base, dtype=np.uint8).reshape(len(ks), 96)[idx].copy()
    few = [0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % O.R, 3, O.R - 3, (1 << 200) + 12345]
    pick = rng.integers(0, len(few), n)
    pick[: n // 2] = 0                                      # half the terms share ONE scalar
    sc = b"".join(few[j].to_bytes(32, "little") for j in pick)
    d_p = ctx.alloc(n * 96); d_p.upload(pts.tobytes())
    d_s = ctx.alloc(n * 32); d_s.upload(sc)
    ctx.set_param("small_msm", 0)
    outs = []
    for tr, rr in ((0, 0), (1, 0), (1, 1)):
        ctx.set_param("tree_row", tr)
        ctx.set_param("rowcol_row", rr)
        outs.append(ctx.msm_device(d_p, d_s, n))
    assert N.cg1_eq(outs[0], outs[1]) == 1 and N.cg1_eq(outs[0], outs[2]) == 1
    tot = sum(ks[i] * few[j] for i, j in zip(idx, pick)) % O.R
    out = ctypes.create_string_buffer(96)
    N.cg1_to_affine96(out, outs[2])
    ref = O.g1_mul(O.G1_GEN, tot) if tot else None
    assert out.raw == (bytes(96) if ref is None else ref[0].to_bytes(48, "little") + ref[1].to_bytes(48, "little"))
    ctx.close()


@pytest.mark.parametrize("n", [1, 3, 4, 5, 64, 255, 585, 2000])
def test_decompress_on_rows_equals_the_pool(n):
    """cg1_batch_decompress_rows (one DPP row per point, the square-root chain with one limb per lane) against cg1_batch_decompress_pool:
    blobs and affine96 byte for byte -- both y signs, the identity's encodings -- and the same first failing index / status for an
    encoding that is off the curve, has x >= p, or lacks the compression flag."""
    import random
    from curdleproofs_pie_amd import _native as N
    from oracle import bls12_381 as O

    ctx = N.Context(0)
    rng = random.Random(700 + n)
    encs = []
    for i in range(n):
        p = O.g1_mul(O.G1_GEN, rng.randrange(1, O.R))
        if rng.random() < 0.5:
            p = O.g1_neg(p)
        encs.append(O.g1_compress(p))
    if n > 4:
        encs[2] = O.g1_compress(None)
        encs[4] = bytes([0xC0 | 0x1F]) + bytes([0xAB]) * 47          # infinity flag with junk below it: still the identity (the wheel's leniency)
    enc = b"".join(encs)

    def both(data, count):
        outs = []
        for gpu in (False, True):
            blobs, aff, bad = ctypes.create_string_buffer(144 * count), ctypes.create_string_buffer(96 * count), ctypes.c_size_t(0)
            if gpu:
                rc = N.cg1_batch_decompress_rows(ctx.handle, data, count, blobs, aff, ctypes.byref(bad))
            else:
                rc = N.cg1_batch_decompress_pool(data, count, blobs, aff, 0, ctypes.byref(bad))
            outs.append((rc, bad.value if rc else 0, blobs.raw if rc == 0 else b"", aff.raw if rc == 0 else b""))
        return outs

    a, b = both(enc, n)
    assert a[0] == N.OK and a == b
    for j in range(min(n, 6)):                                       # and the oracle on a few
        pt = O.g1_decompress(encs[j])
        assert a[3][96 * j: 96 * j + 96] == (bytes(96) if pt is None else pt[0].to_bytes(48, "little") + pt[1].to_bytes(48, "little"))
    if n >= 5:
        x = 5
        while O.fp_sqrt((x ** 3 + 4) % O.P) is not None:
            x += 1
        for pos, bad_enc in ((n - 2, bytes([0x80 | (x >> 376)]) + (x & ((1 << 376) - 1)).to_bytes(47, "big")),      # no y for this x
                             (1, bytes([0x9F]) + bytes([0xFF]) * 47),                                               # x >= p
                             (n // 2, bytes([0x00]) + encs[0][1:])):                                                # compression flag missing
            data = bytearray(enc)
            data[48 * pos: 48 * pos + 48] = bad_enc
            a, b = both(bytes(data), n)
            assert a[0] != N.OK and a[:2] == b[:2], (pos, a[:2], b[:2])
    ctx.close()
