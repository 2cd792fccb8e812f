"""Marshalling of the Python face (no GPU needed): lists of G1Point / Scalar objects <-> contiguous buffers through the C helper
csrc/pyface.c, its pure-Python stand-in, and the per-object normal-form cache (`_a` affine96 / `_k` compressed48) that
MSMAccumulator.accumulate_check (msm_accumulator.py:37-58) and to_compressed_bytes (util.py:27-28) read."""
import ctypes
import random

import pytest

from oracle import bls12_381 as O


@pytest.fixture(scope="module")
def B(native_lib):
    import curdleproofs_pie_amd.py_arkworks_bls12381 as backend

    return backend


def _points(B, rng, n):
    G = B.G1Point()
    prev = B.set_lazy(False)             # computed at once on the host library: projective blobs (Z != 1), like util.get_random_point over the wheel
    try:
        pts = [G * B.Scalar(rng.randint(1, O.R - 1)) for _ in range(n)]
    finally:
        B.set_lazy(prev)
    return pts


def test_helper_is_built(B):
    # build() makes it wherever gcc + Python.h exist (this container, the GPU box); the pure-Python path is the same code otherwise
    from curdleproofs_pie_amd import build as Bd

    Bd.build_pyface(verbose=False)
    import importlib

    assert importlib.util.find_spec("curdleproofs_pie_amd._pyface") is not None


@pytest.mark.parametrize("use_helper", [True, False], ids=["c_helper", "pure_python"])
def test_pack_and_unpack(B, monkeypatch, use_helper):
    if use_helper and B._pyface is None:
        pytest.skip("_pyface not built")
    if not use_helper:
        monkeypatch.setattr(B, "_pyface", None)
    rng = random.Random(5)
    pts = _points(B, rng, 9) + [B.G1Point.identity(), B.G1Point()]
    n = len(pts)
    buf = ctypes.create_string_buffer(144 * n)
    assert tuple(B.pack_points(pts, ctypes.addressof(buf), n)) == (n, 0)
    assert buf.raw == b"".join(p._b for p in pts)
    # the generator and the identity are normal forms; so is anything decoded from bytes
    dec = [B.G1Point.from_compressed_bytes_unchecked(p.to_compressed_bytes()) for p in pts]
    assert tuple(B.pack_points(dec, ctypes.addressof(buf), n)) == (n, 1)
    assert tuple(B.pack_points((), ctypes.addressof(buf), n)) == (0, 1)
    with pytest.raises(ValueError):
        B.pack_points(pts, ctypes.addressof(buf), n - 1)
    with pytest.raises((TypeError, AttributeError)):
        B.pack_points(pts[:2] + [B.Scalar(3)], ctypes.addressof(buf), n)
    back = B.points_from_blobs(buf.raw, n)
    assert back == dec and all(p._a is None and p._k is None for p in back)
    assert B.points_from_blobs(b"", 0) == []

    vals = [0, 1, O.R - 1, rng.randint(0, O.R - 1), 2 ** 255 - 19]
    sc = [B.Scalar(v) for v in vals[:4]] + [vals[4]]                    # Scalars and plain ints (the accumulator's merged scalars)
    sb = ctypes.create_string_buffer(32 * len(sc))
    assert B.pack_scalars(sc, ctypes.addressof(sb), len(sc)) == len(sc)
    assert sb.raw == b"".join(v.to_bytes(32, "little") for v in vals)
    with pytest.raises(ValueError):
        B.pack_scalars(sc, ctypes.addressof(sb), 2)
    with pytest.raises(OverflowError):
        B.pack_scalars([-1], ctypes.addressof(sb), 4)
    with pytest.raises(OverflowError):
        B.pack_scalars([2 ** 256], ctypes.addressof(sb), 4)
    with pytest.raises((TypeError, AttributeError)):
        B.pack_scalars([B.G1Point()], ctypes.addressof(sb), 4)
    # every bit length 0 .. 256 (the C helper reads the 30-bit digits of the int itself): 2^k - 1, 2^k, a random k-bit value
    edge = [0] + [v for k in range(1, 257) for v in ((1 << k) - 1, (1 << (k - 1)), rng.getrandbits(k) | (1 << (k - 1)))]
    eb = ctypes.create_string_buffer(32 * len(edge))
    assert B.pack_scalars(edge, ctypes.addressof(eb), len(edge)) == len(edge)
    assert eb.raw == b"".join(v.to_bytes(32, "little") for v in edge)
    for too_big in (2 ** 256, 2 ** 256 + 1, 2 ** 269, 2 ** 270, 2 ** 300, -(2 ** 40)):
        with pytest.raises(OverflowError):
            B.pack_scalars([1, too_big], ctypes.addressof(eb), 4)

    n1, f1 = B.ident(pts)
    assert n1 == n and B.ident(list(pts)) == (n1, f1) and B.ident(tuple(pts)) == (n1, f1)
    assert B.ident(pts[::-1])[1] != f1 and B.ident(pts[:-1])[1] != f1
    assert B.same_items(pts, tuple(pts)) and not B.same_items(pts, pts[::-1]) and not B.same_items(pts, pts[:-1])
    assert not B.same_items(pts, dec)                                   # equal values, other objects: not the same items


def test_normal_form_cache(B, native_lib):
    N = native_lib
    rng = random.Random(6)
    pts = _points(B, rng, 20) + [B.G1Point.identity()]
    assert all(p._a is None and p._k is None for p in pts)
    want_k = []
    for p in pts:
        out = ctypes.create_string_buffer(48)
        N.cg1_compress(out, p._b)
        want_k.append(out.raw)
    B.ensure_normalised(pts[:7])
    assert all(p._a is not None for p in pts[:7]) and all(p._a is None for p in pts[7:])
    B.ensure_normalised(pts)                                             # only the 14 others are normalised now
    for p, k in zip(pts, want_k):
        assert p._k == k == bytes(p.to_compressed_bytes())
        q = O.g1_decompress(k)
        assert p._a == (bytes(96) if q is None else q[0].to_bytes(48, "little") + q[1].to_bytes(48, "little"))
    assert B.points_to_affine96(pts) == b"".join(p._a for p in pts)
    assert B.points_to_compressed(pts) == want_k
    # to_compressed_bytes alone fills only the key
    p = _points(B, rng, 1)[0]
    k77 = O.g1_compress(O.g1_decompress(bytes(p.to_compressed_bytes())))
    p = B.G1Point._from_blob(p._b)
    assert p._k is None and bytes(p.to_compressed_bytes()) == k77 and p._k is not None and p._a is None
    B.ensure_normalised([p])
    assert p._a == b"".join(c.to_bytes(48, "little") for c in O.g1_decompress(k77))
    # a deferred value leaves its evaluation with blob (Z = 1), affine96 record and encoding
    p = B.G1Point() * B.Scalar(77)
    if B.lazy_enabled():
        assert p._blob is None and p._t is not None
    assert bytes(p.to_compressed_bytes()) == O.g1_compress(O.g1_mul(O.G1_GEN, 77)) and p._k is not None and p._t is None
    B.ensure_normalised([p])
    assert p._a == b"".join(c.to_bytes(48, "little") for c in O.g1_mul(O.G1_GEN, 77))
    # values stay immutable from the outside
    with pytest.raises(AttributeError):
        p._a = None


def test_batch_normalize_entry_point(native_lib):
    N = native_lib
    rng = random.Random(7)
    blobs = []
    for i in range(33):
        b = ctypes.create_string_buffer(144)
        N.cg1_generator(b)
        out = ctypes.create_string_buffer(144)
        N.cg1_mul(out, b.raw, rng.randint(1, O.R - 1).to_bytes(32, "little"))
        blobs.append(out.raw)
    ident = ctypes.create_string_buffer(144)
    N.cg1_identity(ident)
    blobs[11] = ident.raw
    raw = b"".join(blobs)
    n = len(blobs)
    aff, cmp_ = ctypes.create_string_buffer(96 * n), ctypes.create_string_buffer(48 * n)
    assert N.cg1_batch_normalize(raw, n, aff, cmp_) == N.OK
    a2, c2 = ctypes.create_string_buffer(96 * n), ctypes.create_string_buffer(48 * n)
    N.cg1_batch_to_affine96(a2, raw, n)
    N.cg1_batch_compress(c2, raw, n)
    assert aff.raw == a2.raw and cmp_.raw == c2.raw
    # either output may be NULL; n = 0 is fine
    a3 = ctypes.create_string_buffer(96 * n)
    assert N.cg1_batch_normalize(raw, n, a3, None) == N.OK and a3.raw == aff.raw
    c3 = ctypes.create_string_buffer(48 * n)
    assert N.cg1_batch_normalize(raw, n, None, c3) == N.OK and c3.raw == cmp_.raw
    assert N.cg1_batch_normalize(None, 0, None, None) == N.OK
    # and back: affine96 -> blobs
    back = ctypes.create_string_buffer(144 * n)
    assert N.cg1_batch_from_affine96(back, aff.raw, n) == N.OK
    for i in range(n):
        assert N.cg1_eq(back.raw[144 * i: 144 * i + 144], blobs[i]) == 1
