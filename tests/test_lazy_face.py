"""Deferred evaluation of the G1Point operators (curdleproofs_pie_amd/py_arkworks_bls12381.py, csrc/lazy_host.cpp) against the CPU oracle
(oracle/py_arkworks_shim.py: the reference's value classes over pure-Python big integers): the reference's callers use the operators one
element at a time (ipa.py:142-146, same_msm.py:122-126, curdleproofs.py:310-311, util.py:35-36); whatever they ask for -- bytes,
comparisons -- must come back exactly as the wheel would give it, with the evaluation deferred or not, for points of G1 and for
points outside it (the reference decodes unchecked).  No GPU needed: operator batches of this size run on the host's worker pool."""
import ctypes
import random
import threading

import pytest

from oracle import bls12_381 as O
from oracle import py_arkworks_shim as S


@pytest.fixture(scope="module")
def B(native_lib):
    import curdleproofs_pie_amd.py_arkworks_bls12381 as backend

    return backend


T3 = (0, 2)                              # a point of order 3 on E(Fp): on the curve, outside G1 (tests/test_torsion.py)


def _enc(pt):
    return O.g1_compress(pt)


def _imul(pt, k):
    """k * pt for a plain integer k (O.g1_mul reduces k mod r first, which is `Scalar`'s doing, not the group's: outside G1 it matters)."""
    acc = O.JAC_INF
    j = O.jac_from_affine(pt)
    for bit in bin(k)[2:] if k else "":
        acc = O.jac_double(acc)
        if bit == "1":
            acc = O.jac_add(acc, j)
    return O.jac_to_affine(acc)


def test_jacobi_symbol_is_eulers_criterion(native_lib):
    N = native_lib
    rng = random.Random(11)
    vals = [0, 1, 2, 3, 4, O.P - 1, O.P - 2, (O.P - 1) // 2, (O.P + 1) // 2, 2 ** 380, 2 ** 62, 2 ** 62 - 1, 2 ** 124 + 1]
    vals += [rng.randrange(O.P) for _ in range(3000)]
    vals += [rng.randrange(1, 2 ** k) for k in range(1, 381, 3)]
    for v in vals:
        e = pow(v, (O.P - 1) // 2, O.P)
        want = 0 if v == 0 else (1 if e == 1 else -1)
        assert N.cg1_fp_jacobi(v.to_bytes(48, "little")) == want, hex(v)
    assert N.cg1_fp_jacobi(O.P.to_bytes(48, "little")) == 2


def test_validation_without_the_square_root(native_lib):
    """cg1_validate_compressed decides exactly what the decoder decides (oracle.g1_decompress raises / returns), at every flag pattern."""
    N = native_lib
    rng = random.Random(12)
    inf = ctypes.c_int(0)
    cases = []
    for _ in range(400):
        x = rng.randrange(O.P)
        for flags in (0x80, 0xA0):
            cases.append(bytes([flags | (x >> 376)]) + (x & ((1 << 376) - 1)).to_bytes(47, "big"))
    cases += [_enc(O.g1_mul(O.G1_GEN, k)) for k in (1, 2, 3, 99, O.R - 1)]
    cases += [bytes([0xC0]) + bytes(47), bytes([0xE0]) + bytes(47), bytes([0xC0]) + bytes(46) + b"\x01", bytes(48), bytes([0x20]) + bytes(47)]
    cases += [bytes([0x80 | (O.P >> 376)]) + (O.P & ((1 << 376) - 1)).to_bytes(47, "big")]           # x = p
    cases += [_enc(T3)]
    n_ok = 0
    for e in cases:
        try:
            pt = O.g1_decompress(e)
            want = (0, pt is None)
        except ValueError:
            want = None
        rc = N.cg1_validate_compressed(e, ctypes.byref(inf))
        if want is None:
            assert rc in (N.ERR_ENCODING, N.ERR_NOT_ON_CURVE), e.hex()
        else:
            n_ok += 1
            assert rc == N.OK and bool(inf.value) == want[1], e.hex()
    assert 300 < n_ok < len(cases) - 300


def test_subgroup_test_by_endomorphism(native_lib):
    N = native_lib
    rng = random.Random(13)
    pts, want = [], []
    for i in range(24):
        p = O.g1_mul(O.G1_GEN, rng.randrange(1, O.R))
        if i % 3 == 1:
            p = O.g1_add(p, T3)
        elif i % 3 == 2:
            p = O.g1_add(p, O.g1_neg(T3))
        pts.append(p)
        want.append(1 if O.g1_in_subgroup(p) else 0)
    pts += [T3, None]
    want += [0, 1]
    assert want.count(0) == 17
    raw = b"".join(bytes(96) if p is None else p[0].to_bytes(48, "little") + p[1].to_bytes(48, "little") for p in pts)
    flags = ctypes.create_string_buffer(len(pts))
    assert N.cg1_batch_subgroup_pool(raw, len(pts), flags, 0) == N.OK
    assert list(flags.raw) == want


def test_lincomb_batch_pool_against_the_group_law(native_lib):
    N = native_lib
    rng = random.Random(14)
    bases = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(9)] + [None, O.g1_add(O.g1_mul(O.G1_GEN, 5), T3)]
    raw = b"".join(bytes(96) if p is None else p[0].to_bytes(48, "little") + p[1].to_bytes(48, "little") for p in bases)
    offsets, tb, sc, want = [0], [], [], []
    for j in range(40):
        k = rng.choice([0, 1, 1, 2, 3, 5, 17])
        acc = None
        for _ in range(k):
            i = rng.randrange(len(bases))
            neg = rng.random() < 0.3
            s = rng.choice([0, 1, 1, 2, rng.randrange(O.R), rng.randrange(O.R), 2 ** 255 + 12345, 2 ** 256 - 1])
            tb.append(i | (0x80000000 if neg else 0))
            sc.append(s)
            term = _imul(O.g1_neg(bases[i]) if neg else bases[i], s)
            acc = O.g1_add(acc, term)
        offsets.append(len(tb))
        want.append(acc)
    n_out = len(want)
    offs = (ctypes.c_uint32 * (n_out + 1))(*offsets)
    tba = (ctypes.c_uint32 * len(tb))(*tb)
    scb = b"".join(s.to_bytes(32, "little") for s in sc)
    ob, oa, ok = (ctypes.create_string_buffer(144 * n_out), ctypes.create_string_buffer(96 * n_out), ctypes.create_string_buffer(48 * n_out))
    used = ctypes.c_int(0)
    assert N.cg1_lincomb_batch(None, raw, len(bases), offs, n_out, tba, scb, 1, ob, oa, ok, ctypes.byref(used)) == N.OK and used.value == 1
    for j, w in enumerate(want):
        assert ok.raw[48 * j: 48 * j + 48] == O.g1_compress(w), j
        assert oa.raw[96 * j: 96 * j + 96] == (bytes(96) if w is None else w[0].to_bytes(48, "little") + w[1].to_bytes(48, "little"))
        out = ctypes.create_string_buffer(48)
        N.cg1_compress(out, ob.raw[144 * j: 144 * j + 144])
        assert out.raw == O.g1_compress(w)
    # a bad base index is refused, so is the GPU path without a context
    bad = (ctypes.c_uint32 * len(tb))(*([len(bases)] + tb[1:]))
    assert N.cg1_lincomb_batch(None, raw, len(bases), offs, n_out, bad, scb, 1, ob, oa, ok, None) == N.ERR_ARG
    assert N.cg1_lincomb_batch(None, raw, len(bases), offs, n_out, tba, scb, 2, ob, oa, ok, None) == N.ERR_HIP


class _Program:
    """A random straight-line program over the G1Point operators, run on two backends side by side."""

    def __init__(self, rng, n_ops, torsion):
        self.rng, self.n_ops, self.torsion = rng, n_ops, torsion

    def run(self, B, S):
        rng = self.rng
        mine, ref = [B.G1Point(), B.G1Point.identity()], [S.G1Point(), S.G1Point.identity()]
        seeds = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(4)]
        if self.torsion:
            seeds += [O.g1_add(seeds[0], T3), T3, O.g1_add(O.g1_mul(O.G1_GEN, 77), O.g1_neg(T3))]
        for p in seeds:
            e = _enc(p)
            mine.append(B.G1Point.from_compressed_bytes_unchecked(e))
            ref.append(S.G1Point.from_compressed_bytes_unchecked(e))
        checks = 0
        for step in range(self.n_ops):
            op = rng.choice(["add", "add", "sub", "neg", "mul", "mul", "mul", "cmp", "eq", "dec", "small"])
            i, j = rng.randrange(len(mine)), rng.randrange(len(mine))
            if op == "add":
                mine.append(mine[i] + mine[j]); ref.append(ref[i] + ref[j])
            elif op == "sub":
                mine.append(mine[i] - mine[j]); ref.append(ref[i] - ref[j])
            elif op == "neg":
                mine.append(-mine[i]); ref.append(-ref[i])
            elif op == "mul":
                k = rng.choice([0, 1, 2, 3, O.R - 1, rng.randrange(O.R), rng.randrange(O.R)])
                mine.append(mine[i] * B.Scalar(k)); ref.append(ref[i] * S.Scalar(k))
            elif op == "small":
                k = rng.randrange(1, 9)
                mine.append(B.Scalar(k) * mine[i]); ref.append(S.Scalar(k) * ref[i])
            elif op == "cmp":
                assert bytes(mine[i].to_compressed_bytes()) == bytes(ref[i].to_compressed_bytes()), (step, i)
                checks += 1
            elif op == "eq":
                assert (mine[i] == mine[j]) == (ref[i] == ref[j]) and (mine[i] != mine[j]) == (ref[i] != ref[j]), (step, i, j)
                checks += 1
            elif op == "dec":
                e = bytes(ref[i].to_compressed_bytes())
                mine.append(B.G1Point.from_compressed_bytes_unchecked(e)); ref.append(S.G1Point.from_compressed_bytes_unchecked(e))
        for a, b in zip(mine, ref):
            assert str(a) == str(b)
        return checks


@pytest.mark.parametrize("torsion", [False, True], ids=["g1_only", "with_points_outside_g1"])
@pytest.mark.parametrize("lazy", [True, False], ids=["deferred", "eager"])
def test_random_programs_match_the_oracle(B, lazy, torsion):
    prev = B.set_lazy(lazy)
    try:
        before = dict(B.stats)
        for seed in range(6):
            _Program(random.Random(1000 * seed + (7 if torsion else 0)), 60, torsion).run(B, S)
        if lazy:
            assert B.stats["flushes"] > before["flushes"] and B.stats["flushed_values"] > B.stats["flushes"]      # values are evaluated in groups
            if torsion:
                assert B.stats["subgroup_tests"] > before["subgroup_tests"]       # products of products asked whether their bases are in G1
        else:
            assert B.stats["flushes"] == before["flushes"]
    finally:
        B.set_lazy(prev)


def test_products_over_a_base_outside_g1_are_not_folded(B):
    """(P * a) * b == P * (a b mod r) only for P in G1: over P + T3 the wheel multiplies twice, and so must the deferred value."""
    a, b = 2, (O.R + 1) // 2                                  # a b = r + 1: 1 mod r, but r + 1 = 2 (mod 3) as the integer it is [r = 1 mod 3]
    assert a * b % O.R == 1 and (a * b) % 3 == 2
    P = O.g1_add(O.g1_mul(O.G1_GEN, 1234567), T3)
    want = O.g1_mul(O.g1_mul(P, a), b)
    assert want != P                                          # folding the coefficients mod r would give the wrong point
    for lazy in (True, False):
        prev = B.set_lazy(lazy)
        try:
            p = B.G1Point.from_compressed_bytes_unchecked(_enc(P))
            q = (p * B.Scalar(a)) * B.Scalar(b)
            assert bytes(q.to_compressed_bytes()) == _enc(want)
            g = B.G1Point() + p                                # a sum with a leaf outside G1, then scaled twice
            h = (g * B.Scalar(a)) * B.Scalar(b)
            assert bytes(h.to_compressed_bytes()) == _enc(O.g1_mul(O.g1_mul(O.g1_add(O.G1_GEN, P), a), b))
            assert bytes((-(p * B.Scalar(a))).to_compressed_bytes()) == _enc(O.g1_neg(O.g1_mul(P, a)))
            assert p._sg is (False if lazy else None)
        finally:
            B.set_lazy(prev)


def test_decoding_raises_where_the_wheel_raises(B):
    good = _enc(O.g1_mul(O.G1_GEN, 5))
    p = B.G1Point.from_compressed_bytes_unchecked(good)
    if B.lazy_enabled():
        assert p._blob is None and p._k == good             # validated, y not computed
    assert bytes(p.to_compressed_bytes()) == good and p == B.G1Point() * B.Scalar(5)
    x = 1
    while O.fp_sqrt((x ** 3 + 4) % O.P) is not None:
        x += 1
    off_curve = bytes([0x80]) + x.to_bytes(47, "big")
    for bad in (off_curve, bytes(48), good[:47], bytes([0x80 | (O.P >> 376)]) + (O.P & ((1 << 376) - 1)).to_bytes(47, "big")):
        with pytest.raises(ValueError):
            B.G1Point.from_compressed_bytes_unchecked(bad)
        with pytest.raises(ValueError):
            B.G1Point.from_compressed_bytes(bad)
    with pytest.raises(ValueError):
        B.G1Point.from_compressed_bytes(_enc(T3))           # on the curve, outside G1: only the checked decoder refuses it
    t = B.G1Point.from_compressed_bytes_unchecked(_enc(T3))
    assert str(t + t + t) == str(B.G1Point.identity())
    z = B.G1Point.from_compressed_bytes_unchecked(bytes([0xC0]) + bytes(46) + b"\x07")       # infinity flag: the identity whatever follows
    assert z == B.G1Point.identity() and bytes(z.to_compressed_bytes()) == bytes([0xC0]) + bytes(47)


def test_a_flush_takes_the_values_made_after_the_one_asked_for(B):
    if not B.lazy_enabled():
        pytest.skip("deferred evaluation is off")
    G = B.G1Point()
    vals = [G * B.Scalar(k + 2) for k in range(12)]
    later = vals[5] + vals[6]
    before = B.stats["flushes"]
    first = bytes(vals[0].to_compressed_bytes())
    assert B.stats["flushes"] == before + 1
    assert all(v._blob is not None and v._k is not None for v in vals) and later._blob is not None
    assert [bytes(v.to_compressed_bytes()) for v in vals] == [_enc(O.g1_mul(O.G1_GEN, k + 2)) for k in range(12)] and first == _enc(O.g1_mul(O.G1_GEN, 2))
    assert B.stats["flushes"] == before + 1
    # values that die unevaluated cost nothing and leave nothing behind
    tmp = None
    for k in range(300):
        tmp = G * B.Scalar(k + 1)
    del tmp
    tail = G * B.Scalar(4)
    str(tail)
    assert len(B._pending) == 0


def test_operators_from_several_threads(B):
    """Values are immutable and may be used from any thread (the wheel's are): 4 threads build and evaluate shared and private values."""
    G = B.G1Point()
    shared = [G * B.Scalar(k + 3) for k in range(8)]
    want_shared = [_enc(O.g1_mul(O.G1_GEN, k + 3)) for k in range(8)]
    errors = []

    def worker(t):
        try:
            rng = random.Random(t)
            for it in range(60):
                k = rng.randrange(1, 50)
                i = rng.randrange(8)
                v = shared[i] * B.Scalar(k) + G
                if bytes(v.to_compressed_bytes()) != _enc(O.g1_mul(O.G1_GEN, (i + 3) * k + 1)):
                    errors.append((t, it, "value"))
                if bytes(shared[i].to_compressed_bytes()) != want_shared[i]:
                    errors.append((t, it, "shared"))
                d = B.G1Point.from_compressed_bytes_unchecked(want_shared[(i + 1) % 8])
                if not (d + G == shared[(i + 1) % 8] + G):
                    errors.append((t, it, "decoded"))
        except Exception as e:           # noqa: BLE001 -- reported below
            errors.append((t, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert errors == []


def _random_batch(rng, bases, shapes):
    offsets, tb, sc = [0], [], []
    for k in shapes:
        for _ in range(k):
            i = rng.randrange(len(bases))
            tb.append(i | (0x80000000 if rng.random() < 0.3 else 0))
            sc.append(rng.choice([0, 1, 1, 2, rng.randrange(O.R), rng.randrange(O.R), rng.randrange(O.R)]))
        offsets.append(len(tb))
    return offsets, tb, sc


@pytest.mark.gpu
def test_lincomb_batch_on_the_gpu_and_split_between_both(native_lib):
    """cg1_lincomb_batch: the GPU path (one k_msm_small launch for up to 64 combinations, the regime-B chain beyond), the split path
    (large combinations on the GPU while the pool does the small ones) and the pool give the same normalised outputs, byte for byte;
    the pool's are pinned by the oracle above."""
    N = native_lib
    ctx = N.default_context()
    rng = random.Random(15)
    bases = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(40)] + [None, O.g1_add(O.g1_mul(O.G1_GEN, 5), T3)]
    raw = b"".join(bytes(96) if p is None else p[0].to_bytes(48, "little") + p[1].to_bytes(48, "little") for p in bases)
    seen = set()
    for shapes in ([1, 2, 3, 1, 0, 2], [40, 17, 129, 64], [7] * 10 + [1, 2, 3] * 9 + [0], [1] * 56, [300, 5, 1, 1, 2, 0, 9], [3] * 70 + [20] * 3, [5] * 80,
                   [1] * 2100 + [2] * 40 + [9] * 3 + [0]):      # thousands of s * B / A + s * B: the batched scalar-multiplication kernel
        offsets, tb, sc = _random_batch(rng, bases, shapes)
        n_out = len(shapes)
        offs = (ctypes.c_uint32 * (n_out + 1))(*offsets)
        tba = (ctypes.c_uint32 * max(1, len(tb)))(*tb)
        scb = b"".join(s.to_bytes(32, "little") for s in sc) or bytes(32)
        outs = {}
        for path in (1, 2, 0):
            ob, oa, ok = (ctypes.create_string_buffer(144 * n_out), ctypes.create_string_buffer(96 * n_out), ctypes.create_string_buffer(48 * n_out))
            used = ctypes.c_int(0)
            ctx.check(N.cg1_lincomb_batch(ctx.handle, raw, len(bases), offs, n_out, tba, scb, path, ob, oa, ok, ctypes.byref(used)))
            assert used.value == path or path == 0
            seen.add(used.value)
            outs[path] = (ob.raw, oa.raw, ok.raw)
        assert outs[1] == outs[2] == outs[0], shapes
        # the oracle on a sample of the outputs
        for j in rng.sample(range(n_out), min(n_out, 4)):
            acc = None
            for t in range(offsets[j], offsets[j + 1]):
                b = bases[tb[t] & 0x7fffffff]
                acc = O.g1_add(acc, _imul(O.g1_neg(b) if tb[t] >> 31 else b, sc[t]))
            assert outs[0][2][48 * j: 48 * j + 48] == O.g1_compress(acc)
    assert seen == {1, 2, 3}


def _forked_child(q):
    import curdleproofs_pie_amd.py_arkworks_bls12381 as B

    G = B.G1Point()
    vals = [G * B.Scalar(1000 + k) + G for k in range(40)]          # enough work for the pool to be asked for several threads
    q.put([bytes(v.to_compressed_bytes()) for v in vals])


def test_deferred_values_in_a_forked_child(B):
    """multiprocessing workers forked from a parent that already used the native worker pool (the fixture generators work like this):
    the pool's threads do not exist in the child, which starts its own at first use instead of waiting for them forever."""
    import multiprocessing as mp

    G = B.G1Point()
    warm = [G * B.Scalar(7 + k) + G for k in range(40)]
    assert len({bytes(v.to_compressed_bytes()) for v in warm}) == 40    # the parent's pool has run
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    p = ctx.Process(target=_forked_child, args=(q,))
    p.start()
    got = q.get(timeout=120)
    p.join(timeout=30)
    assert p.exitcode == 0
    assert got == [_enc(O.g1_mul(O.G1_GEN, 1001 + k)) for k in range(40)]
