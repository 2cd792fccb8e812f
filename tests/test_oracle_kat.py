"""Pins the CPU oracle (oracle/bls12_381.py and oracle/msm_oracle.c) against every curve-level known answer
the reference's own tests hold, then against the committed golden vectors."""
import random

import pytest

from conftest import raw96
from oracle import bls12_381 as O
from oracle import c_oracle as C

GEN_HEX = "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"  # test_curdleproofs.py:179-180
G99_HEX = "aa10e1055b14a89cc3261699524998732fddc4f30c76c1057eb83732a01416643eb015a932e4080c86f42e485973d240"  # test_curdleproofs.py:236
# not held by the reference: the well-known compression of 2*G (the BLS public key of secret key 2) as an extra pin
G2_HEX = "a572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e"


def test_generator_and_99g_kat():
    assert O.g1_compress(O.G1_GEN).hex() == GEN_HEX
    assert O.g1_compress(O.g1_mul(O.G1_GEN, 99)).hex() == G99_HEX
    assert O.g1_compress(O.g1_mul(O.G1_GEN, 2)).hex() == G2_HEX
    g = raw96(O.G1_GEN)
    assert C.compress(g).hex() == GEN_HEX
    assert C.compress(C.scalar_mul(g, (99).to_bytes(32, "little"))).hex() == G99_HEX
    assert C.compress(C.scalar_mul(g, (2).to_bytes(32, "little"))).hex() == G2_HEX


def test_scalar_rules():
    # test_curdleproofs.py:196-213
    assert O.fr_to_le_bytes(4) == bytes.fromhex("04" + "00" * 31)
    assert O.CURVE_ORDER == 52435875175126190479447740508185965837690552500527637822603658699938581184513
    assert O.fr_from_le_bytes((O.CURVE_ORDER - 1).to_bytes(32, "little")) == O.CURVE_ORDER - 1
    with pytest.raises(ValueError):
        O.fr_from_le_bytes(O.CURVE_ORDER.to_bytes(32, "little"))


def test_group_identities():
    # test_curdleproofs.py:153-176, :241
    g = O.G1_GEN
    dg = O.g1_add(g, g)
    assert O.g1_add(dg, O.g1_neg(g)) == g
    assert O.g1_add(O.g1_neg(g), g) is None
    assert O.g1_mul(g, 4) == O.g1_add(O.g1_add(g, g), O.g1_add(g, g))
    assert O.g1_decompress(O.g1_compress(g), check_subgroup=True) == O.g1_decompress(O.g1_compress(g)) == g
    assert O.g1_mul(g, O.R) is None and O.g1_in_subgroup(g)
    assert O.g1_compress(None) == bytes([0xC0]) + bytes(47) and O.g1_decompress(O.g1_compress(None)) is None


def test_infinity_flag_decodes_to_the_identity_whatever_the_other_bits_say():
    """How the published decoder of the wheel (ark-bls12-381 0.4 read_g1_compressed) treats the infinity flag; C and Python
    restatements agree, and re-serialising gives the canonical encoding the reference hashes (util.py:27-28)."""
    for enc in (bytes([0xC0]) + bytes(47), bytes([0xE0]) + bytes(47), bytes([0xC0]) + bytes(46) + b"\x01", bytes([0xFF]) * 48):
        assert O.g1_decompress(enc) is None and O.g1_decompress(enc, check_subgroup=True) is None
        assert C.decompress(enc) == (0, bytes(96))
        assert O.g1_compress(O.g1_decompress(enc)) == bytes([0xC0]) + bytes(47)


def test_decompress_rejects_bad_encodings():
    for bad in (bytes(48), bytes([0x9F]) + b"\xff" * 47, b"\x80" * 47):
        with pytest.raises(ValueError):
            O.g1_decompress(bad)
    x = 1
    while O.fp_sqrt((x ** 3 + 4) % O.P) is not None:
        x += 1
    enc = bytearray(x.to_bytes(48, "big")); enc[0] |= 0x80
    with pytest.raises(ValueError):
        O.g1_decompress(bytes(enc))


def test_c_oracle_matches_python_oracle():
    rng = random.Random(11)
    pts = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(12)] + [None]
    for n in (0, 1, 2, 13):
        sc = [rng.randint(0, O.R - 1) for _ in range(n)]
        want = O.compute_MSM(pts[:n], sc)
        got = C.compute_msm(b"".join(raw96(p) for p in pts[:n]), b"".join(s.to_bytes(32, "little") for s in sc), n)
        assert got == raw96(want)
        assert C.compress(got) == O.g1_compress(want)
    a, b = pts[0], pts[1]
    assert C.add(raw96(a), raw96(b)) == raw96(O.g1_add(a, b))
    assert C.add(raw96(a), raw96(a)) == raw96(O.g1_mul(a, 2))
    assert C.add(raw96(a), raw96(O.g1_neg(a))) == bytes(96)


def test_c_bucket_oracle_matches_c_naive_oracle():
    rng = random.Random(14)
    pts = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(16)] + [None]
    for n in (0, 1, 17, 1024):
        p96 = b"".join(raw96(pts[rng.randrange(17)]) for _ in range(n))
        s32 = b"".join(rng.choice([0, 1, O.R - 1, rng.randint(0, O.R - 1), rng.randint(0, O.R - 1)]).to_bytes(32, "little") for _ in range(n))
        want = C.compute_msm(p96, s32, n)
        for c in (0, 2, 7, 13):
            assert C.msm_bucket(p96, s32, n, c) == want, (n, c)


def test_naive_and_bucket_oracles_agree():
    rng = random.Random(12)
    pts = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(40)]
    sc = [rng.randint(0, O.R - 1) for _ in range(40)]
    assert O.compute_MSM(pts, sc) == O.compute_MSM_fast(pts, sc, c=5) == O.compute_MSM_fast(pts, sc, c=8)


def test_oracle_reproduces_golden_vectors(golden):
    for case in golden:
        pts = [O.g1_decompress(bytes.fromhex(h)) for h in case["points"]]
        sc = [int.from_bytes(bytes.fromhex(h), "little") for h in case["scalars"]]
        n = len(pts)
        got = C.compute_msm(b"".join(raw96(p) for p in pts), b"".join(s.to_bytes(32, "little") for s in sc), n)
        assert C.compress(got).hex() == case["expected"], case["name"]
        if n <= 7:
            assert O.g1_compress(O.compute_MSM(pts, sc)).hex() == case["expected"], case["name"]


def test_accumulator_restatement():
    # msm_accumulator.py:32-68 semantics: accepts true statements, rejects false ones, ValueError when empty
    rng = random.Random(13)
    pts = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(6)]
    s1 = [rng.randint(1, O.R - 1) for _ in range(6)]
    s2 = [rng.randint(1, O.R - 1) for _ in range(4)]
    acc = O.MSMAccumulator(rng)
    acc.accumulate_check(O.compute_MSM(pts, s1), pts + [None], s1 + [5])
    acc.accumulate_check(O.compute_MSM(pts[:4], s2), pts[:4], s2)
    assert len(acc.base_scalar_map) == 6
    acc.verify()
    bad = O.MSMAccumulator(rng)
    bad.accumulate_check(O.g1_add(O.compute_MSM(pts, s1), O.G1_GEN), pts, s1)
    with pytest.raises(AssertionError):
        bad.verify()
    with pytest.raises(ValueError):
        O.MSMAccumulator(rng).verify()
