"""CPU ORACLE (test infrastructure, NOT product code) -- BLS12-381 G1 / Fr in pure Python big ints.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``curdleproofs_pie_amd``) never does.

What it restates
----------------
* ``compute_MSM``       -- /root/reference/curdleproofs/curdleproofs/msm_accumulator.py:6-12
  (naive  sum_i scalar_i * base_i, one double-and-add scalar multiplication per term, zip semantics)
* ``MSMAccumulator``    -- msm_accumulator.py:32-68 (random-linear-combination batching keyed by the
  48-byte compressed base; one ``random_scalar()`` draw per ``accumulate_check``)
* the ``G1Point`` / ``Scalar`` behaviour of the third-party wheel ``py_arkworks_bls12381`` **0.3.5**
  (pinned at curdleproofs/pyproject.toml:10, curdleproofs/poetry.lock:233-234; its Rust source is NOT
  under /root/reference), as far as the reference pins it:
  stub curdleproofs/py_arkworks_bls12381-stubs/__init__.pyi:5-54 and
  curdleproofs/curdleproofs/test_curdleproofs.py:45-213.
  The published algorithm restated here is the standard one: short-Weierstrass group law on
  y^2 = x^3 + 4 over Fp, the ZCash/IETF 48-byte compressed G1 encoding (bit7 = compressed,
  bit6 = infinity, bit5 = "y is the lexicographically larger root", big-endian x), Fr = integers mod r.

Parity pinning
--------------
The wheel cannot be run in this pipeline (no Rust, no network), so the oracle is pinned by the
reference's own known answers (tests/test_oracle_kat.py):
  * generator compression 97f1d3a7...c6bb   (test_curdleproofs.py:179-180)
  * 99*G compression      aa10e105...d240   (test_curdleproofs.py:233-236)
  * Scalar(4) LE bytes, CURVE_ORDER, reduction/overflow/ValueError rules (test_curdleproofs.py:196-213)
  * the algebraic identities of test_curdleproofs.py:153-176, :241
An MSM result is a mathematically unique group element with a unique canonical encoding, so
bit-exactness of n-term MSM outputs is decidable from these first principles; specific n>1 MSM
output bytes are *not* held by the reference ("parity unpinned" beyond the KATs above).
"""
from __future__ import annotations

import random as _random
from typing import Iterable, List, Optional, Tuple

# --------------------------------------------------------------------------------------
# Public curve constants (BLS12-381).  P, R, B and the generator are public parameters; the
# generator's x is the KAT at test_curdleproofs.py:179 with the three flag bits cleared.
# --------------------------------------------------------------------------------------
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
CURVE_ORDER = 52435875175126190479447740508185965837690552500527637822603658699938581184513  # util.py:7
assert R == CURVE_ORDER
B = 4
GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
assert (GY * GY - GX * GX * GX - B) % P == 0

# Affine points are (x, y) tuples of ints; the identity is None.
Affine = Optional[Tuple[int, int]]
# Jacobian points are (X, Y, Z); identity has Z == 0.
Jac = Tuple[int, int, int]
JAC_INF: Jac = (1, 1, 0)
G1_GEN: Affine = (GX, GY)


# ------------------------------------------------------------------ field helpers
def fp_inv(a: int) -> int:
    return pow(a, P - 2, P)


def fp_sqrt(a: int) -> Optional[int]:
    """p = 3 mod 4, so a^((p+1)/4) is a root iff a is a QR."""
    s = pow(a, (P + 1) // 4, P)
    return s if (s * s - a) % P == 0 else None


# ------------------------------------------------------------------ group law (Jacobian)
def jac_from_affine(pt: Affine) -> Jac:
    return JAC_INF if pt is None else (pt[0], pt[1], 1)


def jac_to_affine(pt: Jac) -> Affine:
    X, Y, Z = pt
    if Z % P == 0:
        return None
    zi = fp_inv(Z)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def jac_double(pt: Jac) -> Jac:
    X, Y, Z = pt
    if Z == 0 or Y == 0:
        return JAC_INF
    A = X * X % P
    Bq = Y * Y % P
    C = Bq * Bq % P
    D = 2 * ((X + Bq) * (X + Bq) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def jac_add(p1: Jac, p2: Jac) -> Jac:
    X1, Y1, Z1 = p1
    X2, Y2, Z2 = p2
    if Z1 == 0:
        return p2
    if Z2 == 0:
        return p1
    Z1Z1 = Z1 * Z1 % P
    Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P
    U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P
    S2 = Y2 * Z1 * Z1Z1 % P
    if U1 == U2:
        if S1 == S2:
            return jac_double(p1)
        return JAC_INF
    H = (U2 - U1) % P
    Rr = (S2 - S1) % P
    HH = H * H % P
    HHH = H * HH % P
    V = U1 * HH % P
    X3 = (Rr * Rr - HHH - 2 * V) % P
    Y3 = (Rr * (V - X3) - S1 * HHH) % P
    Z3 = Z1 * Z2 * H % P
    return (X3, Y3, Z3)


def jac_neg(pt: Jac) -> Jac:
    return (pt[0], (-pt[1]) % P, pt[2])


def jac_mul(pt: Jac, k: int) -> Jac:
    """Double-and-add, MSB first (the algorithm class the wheel's `G1 * Scalar` uses)."""
    k %= R
    acc = JAC_INF
    for bit in bin(k)[2:] if k else "":
        acc = jac_double(acc)
        if bit == "1":
            acc = jac_add(acc, pt)
    return acc


def jac_eq(p1: Jac, p2: Jac) -> bool:
    return jac_to_affine(p1) == jac_to_affine(p2)


# ------------------------------------------------------------------ affine convenience
def g1_add(a: Affine, b: Affine) -> Affine:
    return jac_to_affine(jac_add(jac_from_affine(a), jac_from_affine(b)))


def g1_neg(a: Affine) -> Affine:
    return None if a is None else (a[0], (-a[1]) % P)


def g1_mul(a: Affine, k: int) -> Affine:
    return jac_to_affine(jac_mul(jac_from_affine(a), k))


def g1_is_on_curve(a: Affine) -> bool:
    if a is None:
        return True
    x, y = a
    return 0 <= x < P and 0 <= y < P and (y * y - x * x * x - B) % P == 0


def g1_in_subgroup(a: Affine) -> bool:
    if a is None:
        return True
    # multiply by the group order without reducing the scalar mod r
    acc = JAC_INF
    base = jac_from_affine(a)
    for bit in bin(R)[2:]:
        acc = jac_double(acc)
        if bit == "1":
            acc = jac_add(acc, base)
    return acc[2] % P == 0


# ------------------------------------------------------------------ 48-byte compressed encoding
def g1_compress(a: Affine) -> bytes:
    """ZCash-format compression (what `G1Point.to_compressed_bytes` returns; util.py:27-28)."""
    if a is None:
        return bytes([0xC0]) + bytes(47)
    x, y = a
    flags = 0x80
    if y > (P - 1) // 2:
        flags |= 0x20
    out = bytearray(x.to_bytes(48, "big"))
    out[0] |= flags
    return bytes(out)


def g1_decompress(data: bytes, check_subgroup: bool = False) -> Affine:
    """`from_compressed_bytes_unchecked` (util.py:35-36) / `from_compressed_bytes` (checked).

    Raises ValueError on any malformed encoding (the wheel raises ValueError, test_curdleproofs.py:211-213).
    """
    data = bytes(data)
    if len(data) != 48:
        raise ValueError("G1 compressed encoding must be 48 bytes")
    flags = data[0]
    compressed = bool(flags & 0x80)
    infinity = bool(flags & 0x40)
    largest = bool(flags & 0x20)
    if not compressed:
        raise ValueError("compression flag not set")
    x = int.from_bytes(bytes([flags & 0x1F]) + data[1:], "big")
    if infinity:
        # The identity as soon as the infinity flag is set, whatever the sign flag and the other 381 bits say: that is how the
        # wheel's decoder (ark-bls12-381 0.4 `read_g1_compressed`) is published; it re-serialises the point canonically wherever
        # the reference hashes it (util.py:27-28).  Not pinned by a reference vector (the wheel cannot run here).
        return None
    if x >= P:
        raise ValueError("x not canonical")
    y = fp_sqrt((x * x * x + B) % P)
    if y is None:
        raise ValueError("x is not on the curve")
    if (y > (P - 1) // 2) != largest:
        y = P - y
    pt = (x, y)
    if check_subgroup and not g1_in_subgroup(pt):
        raise ValueError("point not in the prime-order subgroup")
    return pt


# ------------------------------------------------------------------ Fr helpers
def fr_from_le_bytes(data: bytes) -> int:
    data = bytes(data)
    if len(data) != 32:
        raise ValueError("Fr encoding must be 32 bytes")
    v = int.from_bytes(data, "little")
    if v >= R:
        raise ValueError("serialised data seems to be invalid")  # test_curdleproofs.py:210-213
    return v


def fr_to_le_bytes(v: int) -> bytes:
    return (v % R).to_bytes(32, "little")


def random_scalar(rng=_random) -> int:
    """util.py:21-24: uniform in [1, r-1] from Python's `random` module."""
    return rng.randint(1, CURVE_ORDER - 1)


# ------------------------------------------------------------------ the hot path
def compute_MSM(bases: Iterable[Affine], scalars: Iterable[int]) -> Affine:
    """msm_accumulator.py:6-12.  `zip` semantics: truncates to the shorter iterable."""
    current = JAC_INF  # :9
    for base, scalar in zip(bases, scalars):  # :10
        current = jac_add(current, jac_mul(jac_from_affine(base), scalar))  # :11
    return jac_to_affine(current)  # :12


def compute_MSM_fast(bases: List[Affine], scalars: List[int], c: int = 8) -> Affine:
    """Same group element as compute_MSM, via a simple bucket method (used by tests to cross-check
    the naive loop at sizes where the naive loop would take minutes).  Not a reference restatement."""
    pairs = list(zip(bases, scalars))
    nwin = (255 + c - 1) // c
    total = JAC_INF
    for w in reversed(range(nwin)):
        for _ in range(c):
            total = jac_double(total)
        buckets: List[Jac] = [JAC_INF] * (1 << c)
        for base, s in pairs:
            d = ((s % R) >> (w * c)) & ((1 << c) - 1)
            if d and base is not None:
                buckets[d] = jac_add(buckets[d], jac_from_affine(base))
        run = JAC_INF
        acc = JAC_INF
        for d in range((1 << c) - 1, 0, -1):
            run = jac_add(run, buckets[d])
            acc = jac_add(acc, run)
        total = jac_add(total, acc)
    return jac_to_affine(total)


class MSMAccumulator:
    """msm_accumulator.py:32-68 restated over affine tuples / int scalars."""

    def __init__(self, rng=_random) -> None:
        self.A_c: Jac = JAC_INF  # :34
        self.base_scalar_map = {}  # :35
        self._rng = rng

    def accumulate_check(self, C: Affine, bases: Iterable[Affine], scalars: Iterable[int]) -> None:
        random_factor = random_scalar(self._rng)  # :43
        self.A_c = jac_add(self.A_c, jac_mul(jac_from_affine(C), random_factor))  # :45
        for base, scalar in zip(bases, scalars):  # :47
            if base is None:  # :49-50
                continue
            key = g1_compress(base)  # :54
            self.base_scalar_map[key] = (self.base_scalar_map.get(key, 0) + random_factor * scalar) % R  # :56-58

    def verify(self) -> None:
        if not self.base_scalar_map:
            # `zip(*{}.items())` unpacks into nothing -> ValueError in the reference (:63)
            raise ValueError("not enough values to unpack (expected 2, got 0)")
        keys, scalars = map(list, zip(*self.base_scalar_map.items()))  # :63
        computed = compute_MSM([g1_decompress(k) for k in keys], scalars)  # :64-67
        assert computed == jac_to_affine(self.A_c)  # :68
