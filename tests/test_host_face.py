"""Drop-in surface of the G1Point / Scalar backend (host side of libcurdle_g1.so; no GPU needed).
Mirrors the reference's own backend tests: curdleproofs/curdleproofs/test_curdleproofs.py:45-241."""
import random

import pytest

from oracle import bls12_381 as O


@pytest.fixture(scope="module")
def B(native_lib):
    import curdleproofs_pie_amd.py_arkworks_bls12381 as backend

    return backend


def test_api_snapshot(B):
    # test_curdleproofs.py:45-128: the exact dir() of both classes
    assert dir(B.G1Point) == B._G1_DIR and dir(B.Scalar) == B._SCALAR_DIR
    for name in B._G1_DIR:
        assert hasattr(B.G1Point, name), name
    for name in B._SCALAR_DIR:
        assert hasattr(B.Scalar, name), name


def test_g1points(B):
    # test_curdleproofs.py:132-191
    G1Point, Scalar = B.G1Point, B.Scalar
    gen = G1Point()
    identity = G1Point.identity()
    assert gen == gen and gen != identity
    double_gen = gen + gen
    assert double_gen - gen == gen
    assert -gen + gen == identity
    assert gen * Scalar(4) == gen + gen + gen + gen
    cb = gen.to_compressed_bytes()
    assert G1Point.from_compressed_bytes(cb) == G1Point.from_compressed_bytes_unchecked(cb) == gen
    assert str(gen) == "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
    assert bytes(gen.to_compressed_bytes()) == bytes.fromhex(str(gen))
    with pytest.raises(TypeError):
        {gen * Scalar(4): True}
    # test_curdleproofs.py:233-236 (asserted here, unlike upstream)
    assert bytes((gen * Scalar(99)).to_compressed_bytes()).hex() == "aa10e1055b14a89cc3261699524998732fddc4f30c76c1057eb83732a01416643eb015a932e4080c86f42e485973d240"
    assert str(identity) == "c0" + "00" * 47


def test_scalar(B):
    # test_curdleproofs.py:194-230
    Scalar, R = B.Scalar, B.CURVE_ORDER
    assert bytes(Scalar(4).to_le_bytes()) == bytes.fromhex("04" + "00" * 31)
    assert int(Scalar(R - 1)) == R - 1 and int(Scalar(R)) == 0
    assert int(Scalar(2 ** 256)) == 2 ** 256 % R and int(Scalar(2 ** 257)) == 2 ** 257 % R
    Scalar.from_le_bytes((R - 1).to_bytes(32, "little"))
    with pytest.raises(ValueError):
        Scalar.from_le_bytes(R.to_bytes(32, "little"))
    a, b = Scalar(1234567), Scalar(R - 5)
    assert int(a + b) == (1234567 + R - 5) % R and int(a - b) == (1234567 + 5) % R and int(-a) == R - 1234567
    assert int(a * b) == 1234567 * (R - 5) % R and a.inverse() * a == Scalar(1) and (a / a) == Scalar(1)
    assert int(a.square()) == 1234567 ** 2 % R and int(a.pow(5)) == pow(1234567, 5, R)
    assert Scalar(0).is_zero() and not a.is_zero() and Scalar(0).inverse() == Scalar(0)
    assert sum([a, b], Scalar(0)) == a + b


def test_util_helpers(B, native_lib):
    from curdleproofs_pie_amd import util as U

    random.seed(7)
    p = U.get_random_point()
    assert p + U.G1 - p == U.G1                                     # test_curdleproofs.py:239-241
    assert U.g1_is_inf(U.Z1) and not U.g1_is_inf(U.G1)
    s = U.random_scalar()
    assert s.inverse() * s == B.Scalar(1)
    assert U.point_projective_from_bytes(U.point_projective_to_bytes(p)) == p
    assert B.Scalar.from_le_bytes(U.field_to_bytes(s)) == s
    # seeded draws follow Python's global `random` exactly like util.py:21-24
    random.seed(99); a = int(U.random_scalar()); random.seed(99)
    assert a == random.randint(1, B.CURVE_ORDER - 1)


def test_against_oracle(B):
    rng = random.Random(5)
    G1Point, Scalar = B.G1Point, B.Scalar
    g = G1Point()
    for _ in range(10):
        k, j = rng.randrange(O.R), rng.randrange(O.R)
        a, b = g * Scalar(k), g * Scalar(j)
        assert a.to_compressed_bytes() == O.g1_compress(O.g1_mul(O.G1_GEN, k))
        assert (a + b).to_compressed_bytes() == O.g1_compress(O.g1_mul(O.G1_GEN, (k + j) % O.R))
        assert (a - b).to_compressed_bytes() == O.g1_compress(O.g1_mul(O.G1_GEN, (k - j) % O.R))
        assert G1Point.from_compressed_bytes((-a).to_compressed_bytes()) == -a
    assert g * Scalar(O.R - 1) + g == G1Point.identity()
    pts = [g * Scalar(rng.randrange(O.R)) for _ in range(5)] + [G1Point.identity()]
    assert B.points_to_compressed(pts) == [p.to_compressed_bytes() for p in pts]


def test_infinity_flag_decodes_to_the_identity(B):
    """any encoding with the infinity flag is the identity (as the oracle restates the wheel's decoder), re-encoded canonically"""
    for enc in (bytes([0xC0]) + bytes(47), bytes([0xE0]) + bytes(47), bytes([0xC0]) + bytes(46) + b"\x01", bytes([0xFF]) * 48):
        for f in (B.G1Point.from_compressed_bytes_unchecked, B.G1Point.from_compressed_bytes):
            p = f(enc)
            assert p == B.G1Point.identity() and bytes(p.to_compressed_bytes()) == O.g1_compress(O.g1_decompress(enc))


def test_bad_encodings_raise_valueerror(B):
    G1Point = B.G1Point
    for bad in (bytes(48), bytes([0x9F]) + b"\xff" * 47, b"\x80" * 47):
        with pytest.raises(ValueError):
            G1Point.from_compressed_bytes_unchecked(bad)
    x = 1
    while O.fp_sqrt((x ** 3 + 4) % O.P) is not None:
        x += 1
    enc = bytearray(x.to_bytes(48, "big")); enc[0] |= 0x80
    with pytest.raises(ValueError):
        G1Point.from_compressed_bytes_unchecked(bytes(enc))
    x = 1
    while True:  # on the curve but outside the prime-order subgroup: only the checked decoder rejects it
        y = O.fp_sqrt((x ** 3 + 4) % O.P)
        if y is not None and not O.g1_in_subgroup((x, y)):
            break
        x += 1
    enc = bytearray(x.to_bytes(48, "big")); enc[0] |= 0x80
    G1Point.from_compressed_bytes_unchecked(bytes(enc))
    with pytest.raises(ValueError):
        G1Point.from_compressed_bytes(bytes(enc))


def test_scalar_mul_recoding_edges(native_lib):
    """cg1_mul (G1Point * Scalar, the width-5 NAF ladder of host_g1.cpp) against the oracle's double-and-add on scalars that stress the
    recoding: 0, 1, every digit boundary around 15 / 16 / 17 / 31 / 32 / 33, long runs of ones, r - 1, and unreduced 256-bit values
    (the C ABI takes any 32 bytes; the Python face only ever passes values < r)."""
    import ctypes
    import random

    N = native_lib
    rng = random.Random(77)
    g = ctypes.create_string_buffer(144)
    N.cg1_generator(g)
    base = ctypes.create_string_buffer(144)
    N.cg1_mul(base, g.raw, (0xABCDEF12345).to_bytes(32, "little"))          # a projective base (Z != 1)
    base_aff = O.g1_mul(O.G1_GEN, 0xABCDEF12345)
    ks = [0, 1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 48, 49, (1 << 64) - 1, 1 << 64, (1 << 128) + 1, (1 << 255) - 1, (1 << 255), (1 << 256) - 1,
          O.R - 1, O.R, O.R + 1, int("5" * 64, 16), int("a" * 64, 16), int("f0" * 32, 16), int("0f" * 32, 16)]
    ks += [rng.getrandbits(256) for _ in range(40)] + [rng.getrandbits(rng.randrange(1, 257)) for _ in range(40)]
    for k in ks:
        out = ctypes.create_string_buffer(144)
        N.cg1_mul(out, base.raw, k.to_bytes(32, "little"))
        cmp_ = ctypes.create_string_buffer(48)
        N.cg1_compress(cmp_, out.raw)
        assert cmp_.raw == O.g1_compress(O.g1_mul(base_aff, k % O.R)), hex(k)
    ident = ctypes.create_string_buffer(144)
    N.cg1_identity(ident)
    out = ctypes.create_string_buffer(144)
    N.cg1_mul(out, ident.raw, (12345).to_bytes(32, "little"))
    assert N.cg1_is_identity(out.raw) == 1


def test_decoded_points_keep_their_encoding(B):
    """from_compressed_bytes[_unchecked] of a finite point remembers the 48 bytes it came from (they are its compression); the
    identity's encoding is remembered only through the normal path, because the decoder accepts non-canonical infinity encodings."""
    p = B.G1Point() * B.Scalar(424242)
    enc = bytes(p.to_compressed_bytes())
    q = B.G1Point.from_compressed_bytes_unchecked(enc)
    assert q._k == enc and bytes(q.to_compressed_bytes()) == enc and q == p
    q = B.G1Point.from_compressed_bytes(enc)
    assert q._k == enc
    odd_inf = bytes([0xC0 | 0x20]) + bytes(46) + b"\x01"                      # infinity flag + stray bits: still the identity
    z = B.G1Point.from_compressed_bytes_unchecked(odd_inf)
    assert z._k is None and bytes(z.to_compressed_bytes()) == bytes([0xC0]) + bytes(47) and z == B.G1Point.identity()


def test_batch_mul_add_pool_against_oracle():
    """cg1_batch_mul_add_pool (the host pool's path of cg1_batch_mul_add: folds / maps of a few hundred points, ipa.py:142-146,
    curdleproofs.py:310-311, grand_prod.py:64-71) against the oracle's group law: all three patterns, identities, 1 and 3 threads,
    a coordinate >= p refused."""
    import ctypes
    import random

    from conftest import raw96
    from curdleproofs_pie_amd import _native as N
    from oracle import bls12_381 as O

    rng = random.Random(5)
    n = 24
    pts = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(n)]
    pts[0] = None
    add = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(n)]
    add[1] = None
    add[2] = O.g1_neg(pts[2])
    sc = [rng.randint(0, O.R - 1) for _ in range(n)]
    sc[2], sc[5], sc[6] = 1, 0, O.R - 1
    b96, a96 = b"".join(raw96(p) for p in pts), b"".join(raw96(p) for p in add)
    s32 = b"".join(s.to_bytes(32, "little") for s in sc)
    out = ctypes.create_string_buffer(96 * n)
    memo = {}

    def mul(j, k):
        if (j, k) not in memo:
            memo[(j, k)] = O.g1_mul(pts[j], k)
        return memo[(j, k)]

    for threads in (1, 3):
        for nbase, scal, nsc, addend in ((n, s32, n, None), (n, s32[64:96], 1, a96), (7, s32, n, a96), (1, s32[:32 * 9], 9, None)):
            assert N.cg1_batch_mul_add_pool(b96, nbase, scal, nsc, addend, out, n, threads) == 0
            for i in range(n):
                sci = int.from_bytes(scal[32 * (i % nsc): 32 * (i % nsc) + 32], "little")
                want = O.g1_add(None if addend is None else add[i], mul(i % nbase, sci))
                assert out.raw[96 * i: 96 * i + 96] == raw96(want), (threads, nbase, nsc, i)
    assert N.cg1_batch_mul_add_pool(b96, n, s32, n, None, out, 0, 0) == 0
    bad = bytearray(b96)
    bad[96 * 3: 96 * 3 + 48] = b"\xff" * 48                    # x >= p
    assert N.cg1_batch_mul_add_pool(bytes(bad), n, s32, n, None, out, n, 2) != 0
