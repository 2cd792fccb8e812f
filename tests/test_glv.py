"""The endomorphism split k = k1 + k2 * lambda (csrc/glv.h) as the device digit kernels compute it, against Python integers; on the
GPU: an MSM with the split against the same MSM without it, and the refusal outside the subgroup being the CALLER's to make."""
import ctypes
import os
import random

import numpy as np
import pytest

from curdleproofs_pie_amd import _native as N
from oracle import bls12_381 as O

Z = 0xD201000000010000
LAM = Z * Z - 1
R = O.R


def _split(k: int):
    k1 = ctypes.create_string_buffer(16)
    k2 = ctypes.create_string_buffer(16)
    n1, n2 = ctypes.c_int(), ctypes.c_int()
    N.cg1_glv_split(k.to_bytes(32, "little"), k1, k2, ctypes.byref(n1), ctypes.byref(n2))
    a, b = int.from_bytes(k1.raw, "little"), int.from_bytes(k2.raw, "little")
    return (-a if n1.value else a), (-b if n2.value else b)


def test_lambda_is_the_eigenvalue_of_phi():
    assert (LAM * LAM + LAM + 1) % R == 0
    beta = 0x1A0111EA397FE699EC02408663D4DE85AA0D857D89759AD4897D29650FB85F9B409427EB4F49FFFD8BFD00000000AAAC
    g = O.G1_GEN
    p = O.g1_mul(g, 0x1234567)
    assert O.g1_mul(p, LAM) == (p[0] * beta % O.P, p[1])


def test_split_identity_and_bounds():
    rng = random.Random(5)
    edge = [0, 1, 2, R - 1, R, R + 1, (R + 1) // 2, (R + 1) // 2 - 1, (R + 1) // 2 + 1, (1 << 255) - 1, LAM, LAM - 1, LAM + 1, LAM >> 1, (LAM >> 1) + 1,
            LAM * LAM, LAM * (LAM >> 1), R - LAM, R // 2, 1 << 254, (1 << 254) - 1, 1 << 127, (1 << 128) - 1]
    cases = edge + [rng.getrandbits(255) for _ in range(20000)] + [rng.getrandbits(rng.randrange(1, 255)) for _ in range(2000)]
    bound = (LAM + 1) // 2 + 1
    for k in cases:
        k1, k2 = _split(k)
        assert (k1 + k2 * LAM - k) % R == 0, hex(k)
        assert abs(k1) <= bound and abs(k2) <= bound, hex(k)


@pytest.mark.parametrize("glv", [0, 1])
def test_every_window_plan_tiles_the_bit_positions(glv):
    """Uniform and balanced plans, every width: windows start at bit 0, follow each other without gap or overlap, are cmax or cmax - 1
    wide, and cover all 256 (128 with the split) positions -- a balanced plan whose narrow windows alone cover the range (128 positions
    at width 14) once produced a negative first offset."""
    positions = 128 if glv else 256
    for c in list(range(4, 17)) + [-x for x in range(4, 17)]:
        offs = (ctypes.c_int * 80)()
        wid = (ctypes.c_int * 80)()
        nw = N.cg1_plan_describe(c, glv, offs, wid, 80)
        assert nw > 0, c
        assert offs[0] == 0, (c, list(offs[:nw]))
        for w in range(nw):
            assert wid[w] in (abs(c), abs(c) - 1) and wid[w] >= 1
            if w:
                assert offs[w] == offs[w - 1] + wid[w - 1], (c, w)
        assert offs[nw - 1] + wid[nw - 1] >= positions, c
        assert offs[nw - 1] <= positions - 1, c                     # (255 / c integral: the top window holds the recoding carry only)
    assert N.cg1_plan_describe(3, glv, offs, wid, 80) == -1 and N.cg1_plan_describe(17, glv, offs, wid, 80) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("n,c", [(3000, 0), (1 << 14, 0), (1 << 14, 16), (1 << 14, -13), (1 << 14, -14), (1 << 14, 14), (1 << 14, 11), ((1 << 16) + 77, 0), (1 << 18, 0)])
def test_msm_with_the_split_equals_the_msm_without(n, c):
    ctx = N.Context(0)
    rng = np.random.default_rng(n + 100 * abs(c))
    # points of G1: multiples of the generator made on the device, with a few identities and repeats
    m = min(n, 4096)
    gen = O.G1_GEN
    gen96 = gen[0].to_bytes(48, "little") + gen[1].to_bytes(48, "little")
    ks = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    ks[:, 31] &= 0x3F
    base = ctx.batch_mul_add_host(gen96, 1, ks.tobytes(), m, None, m)
    pts = np.frombuffer(base, dtype=np.uint8).reshape(m, 96)
    pts = pts[rng.integers(0, m, n)].copy()
    pts[5] = 0
    pts[n // 2] = 0
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x3F
    sc[0] = 0
    sc[1] = np.frombuffer((R - 1).to_bytes(32, "little"), dtype=np.uint8)
    sc[2] = np.frombuffer(((R + 1) // 2).to_bytes(32, "little"), dtype=np.uint8)
    sc[3] = np.frombuffer((1).to_bytes(32, "little"), dtype=np.uint8)
    sc[4, 16:] = 0                                        # a 128-bit scalar
    d_p = ctx.alloc(n * 96); d_p.upload(pts.tobytes())
    d_s = ctx.alloc(n * 32); d_s.upload(sc.tobytes())
    ctx.set_param("glv", 0)
    want = ctx.msm_device(d_p, d_s, n, window_c=c)
    ctx.set_param("glv", 2)                               # 2: wherever the engine can (1 stops at glv_max_n terms in regime A)
    got = ctx.msm_device(d_p, d_s, n, window_c=c)
    assert N.cg1_eq(got, want) == 1
    if n <= 3000:
        aff = [(int.from_bytes(p[:48].tobytes(), "little"), int.from_bytes(p[48:].tobytes(), "little")) for p in pts]
        aff = [None if a == (0, 0) else a for a in aff]
        ref = O.compute_MSM_fast(aff, [int.from_bytes(s.tobytes(), "little") for s in sc])
        out = ctypes.create_string_buffer(96)
        N.cg1_to_affine96(out, got)
        assert out.raw == (bytes(96) if ref is None else ref[0].to_bytes(48, "little") + ref[1].to_bytes(48, "little"))
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 3, 5, 17, 96, 97, 124, 128, 255, 256, 257, 400, 512, 627, 1000, 1024])
def test_small_kernel_with_the_split(n):
    """k_msm_small over the 2n entries of the split (n <= 1 024) against the same call without it and against the oracle."""
    ctx = N.Context(0)
    rng = np.random.default_rng(1000 + n)
    gen96 = O.G1_GEN[0].to_bytes(48, "little") + O.G1_GEN[1].to_bytes(48, "little")
    ks = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x3F
    pts = np.frombuffer(ctx.batch_mul_add_host(gen96, 1, ks.tobytes(), n, None, n), dtype=np.uint8).reshape(n, 96).copy()
    if n > 4:
        pts[3] = 0
        pts[4] = pts[2]
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x3F
    sc[0] = np.frombuffer((R - 1).to_bytes(32, "little"), dtype=np.uint8)
    if n > 2:
        sc[1] = 0
        sc[2] = sc[4 % n] if n > 4 else sc[2]
    d_p = ctx.alloc(n * 96); d_p.upload(pts.tobytes())
    d_s = ctx.alloc(n * 32); d_s.upload(sc.tobytes())
    ctx.set_param("glv", 0)
    want = ctx.msm_device(d_p, d_s, n)
    assert ctx.last_counts()["accumulate_launches"] == 0          # the single-launch kernel
    ctx.set_param("glv", 1)
    got = ctx.msm_device(d_p, d_s, n)
    assert ctx.last_counts()["accumulate_launches"] == 0
    assert N.cg1_eq(got, want) == 1
    aff = [(int.from_bytes(p[:48].tobytes(), "little"), int.from_bytes(p[48:].tobytes(), "little")) for p in pts]
    aff = [None if a == (0, 0) else a for a in aff]
    ref = O.compute_MSM_fast(aff, [int.from_bytes(s.tobytes(), "little") for s in sc]) if n > 8 else O.compute_MSM(aff, [int.from_bytes(s.tobytes(), "little") for s in sc])
    out = ctypes.create_string_buffer(96)
    N.cg1_to_affine96(out, got)
    assert out.raw == (bytes(96) if ref is None else ref[0].to_bytes(48, "little") + ref[1].to_bytes(48, "little"))
    # several MSMs in one launch (the deferred flush's shape): 5 MSMs over slices of the same input
    if n >= 20:
        offs = [0, n // 5, 2 * n // 5, 3 * n // 5, 4 * n // 5, n]
        ctx.set_param("glv", 0)
        a = ctx.msm_batched_device(d_p, d_s, offs)
        ctx.set_param("glv", 1)
        b = ctx.msm_batched_device(d_p, d_s, offs)
        assert all(N.cg1_eq(x, y) == 1 for x, y in zip(a, b))
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M,n,c", [(200, 70, 0), (100, 627, 0), (100, 627, 6), (100, 627, 8), (300, 33, 0), (513, 40, 0)])
def test_regime_b_with_the_split(M, n, c):
    """M independent MSMs in the regime-B chain (more than one single-launch batch): split against no split, ragged offsets with
    empty MSMs, and a few of them against the oracle."""
    ctx = N.Context(0)
    rng = np.random.default_rng(M * 1000 + n + c)
    tot = M * n
    gen96 = O.G1_GEN[0].to_bytes(48, "little") + O.G1_GEN[1].to_bytes(48, "little")
    m = 512
    ks = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    ks[:, 31] &= 0x3F
    base = np.frombuffer(ctx.batch_mul_add_host(gen96, 1, ks.tobytes(), m, None, m), dtype=np.uint8).reshape(m, 96)
    pts = base[rng.integers(0, m, tot)].copy()
    pts[7] = 0
    sc = rng.integers(0, 256, (tot, 32), dtype=np.uint8)
    sc[:, 31] &= 0x3F
    sc[3] = 0
    cuts = sorted(int(x) for x in rng.integers(0, tot + 1, M - 1))
    offs = [0] + cuts + [tot]
    offs[5] = offs[4]                                       # an empty MSM
    offs = sorted(offs)
    d_p = ctx.alloc(tot * 96); d_p.upload(pts.tobytes())
    d_s = ctx.alloc(tot * 32); d_s.upload(sc.tobytes())
    ctx.set_param("glv", 0)
    a = ctx.msm_batched_device(d_p, d_s, offs, window_c=c)
    ctx.set_param("glv", 1)
    b = ctx.msm_batched_device(d_p, d_s, offs, window_c=c)
    assert len(a) == len(b) == M
    if M >= 256:                                            # the two-chain form (A/B switch, off by default) must agree
        ctx.set_param("batched_split", 1)
        b1 = ctx.msm_batched_device(d_p, d_s, offs, window_c=c)
        ctx.set_param("batched_split", 0)
        assert all(N.cg1_eq(x, y) == 1 for x, y in zip(b, b1))
    assert all(N.cg1_eq(x, y) == 1 for x, y in zip(a, b))
    for j in (0, 4, M - 1):
        lo, hi = offs[j], offs[j + 1]
        aff = [(int.from_bytes(p[:48].tobytes(), "little"), int.from_bytes(p[48:].tobytes(), "little")) for p in pts[lo:hi]]
        aff = [None if q == (0, 0) else q for q in aff]
        ss = [int.from_bytes(s.tobytes(), "little") for s in sc[lo:hi]]
        ref = O.compute_MSM_fast(aff, ss) if hi - lo > 8 else O.compute_MSM(aff, ss)
        out = ctypes.create_string_buffer(96)
        N.cg1_to_affine96(out, b[j])
        assert out.raw == (bytes(96) if ref is None else ref[0].to_bytes(48, "little") + ref[1].to_bytes(48, "little"))
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [6, 60, 600, 3000, 20000])
def test_split_with_bases_that_are_each_others_images(n):
    """P, phi(P), phi^2(P) and their negatives side by side, equal scalars among them: the split's second halves phi(P_i) then COINCIDE with
    other inputs (and with their negatives) inside the same buckets -- the doubling and cancellation cases of the bucket additions --
    and sums like P + phi(P) + phi^2(P) = 0 appear.  Split against no split against the oracle."""
    beta = 0x1A0111EA397FE699EC02408663D4DE85AA0D857D89759AD4897D29650FB85F9B409427EB4F49FFFD8BFD00000000AAAC
    ctx = N.Context(0)
    rng = random.Random(900 + n)
    seeds = [O.g1_mul(O.G1_GEN, rng.randrange(1, R)) for _ in range(3)]
    fam = []
    for p in seeds:
        p1 = (p[0] * beta % O.P, p[1])
        p2 = (p1[0] * beta % O.P, p[1])
        fam += [p, p1, p2, O.g1_neg(p), O.g1_neg(p1), O.g1_neg(p2)]
    few = [rng.randrange(R) for _ in range(3)] + [1, R - 1, LAM, LAM + 1, R - LAM, (R + 1) // 2]
    pts, sc = [], []
    for i in range(n):
        pts.append(fam[i % len(fam)] if i % 7 else rng.choice(fam))
        sc.append(rng.choice(few) if i % 3 else rng.randrange(R))
    raw = b"".join(p[0].to_bytes(48, "little") + p[1].to_bytes(48, "little") for p in pts)
    s32 = b"".join(s.to_bytes(32, "little") for s in sc)
    d_p = ctx.alloc(n * 96); d_p.upload(raw)
    d_s = ctx.alloc(n * 32); d_s.upload(s32)
    ctx.set_param("glv", 0)
    want = ctx.msm_device(d_p, d_s, n)
    ctx.set_param("glv", 2)
    got = ctx.msm_device(d_p, d_s, n)
    assert N.cg1_eq(got, want) == 1
    if n <= 3000:
        ref = O.compute_MSM_fast(pts, sc) if n > 8 else O.compute_MSM(pts, sc)
        out = ctypes.create_string_buffer(96)
        N.cg1_to_affine96(out, got)
        assert out.raw == (bytes(96) if ref is None else ref[0].to_bytes(48, "little") + ref[1].to_bytes(48, "little"))
    if n >= 60:                                             # the same inputs as 12 MSMs of one launch, and as regime B (M = 90)
        for M in (12, 90):
            if M * 2 > n:
                continue
            offs = [n * j // M for j in range(M + 1)]
            ctx.set_param("glv", 0)
            a = ctx.msm_batched_device(d_p, d_s, offs)
            ctx.set_param("glv", 1)
            b = ctx.msm_batched_device(d_p, d_s, offs)
            assert all(N.cg1_eq(x, y) == 1 for x, y in zip(a, b))
    ctx.close()
