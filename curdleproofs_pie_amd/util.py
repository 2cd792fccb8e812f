"""Boundary helpers of /root/reference/curdleproofs/curdleproofs/util.py that sit directly on the
G1Point / Scalar backend and that THIS package calls.  Same names, same behaviour, same
randomness source (Python's global `random`, so `random.seed(k)` reproduces the reference's draw order).  The reference's
pure-Python helpers (invert, scalar_pow, inner_product, BufReader, ...) are not restated: its callers keep using their own
util.py over the injected backend (tests/test_reference_suite.py)."""
from __future__ import annotations

from random import randint
from typing import List

from .py_arkworks_bls12381 import CURVE_ORDER, G1Point, Scalar

G1 = G1Point()            # util.py:9
Z1 = G1Point.identity()   # util.py:11


def g1_is_inf(point: G1Point) -> bool:  # util.py:17-18
    return point == Z1


def random_scalar() -> Scalar:  # util.py:21-24
    return Scalar.from_le_bytes(randint(1, CURVE_ORDER - 1).to_bytes(32, "little"))


def point_projective_to_bytes(point: G1Point) -> bytes:  # util.py:27-28
    return bytes(point.to_compressed_bytes())


def points_projective_to_bytes(points: List[G1Point]) -> List[bytes]:  # util.py:31-32
    from .py_arkworks_bls12381 import points_to_compressed

    return points_to_compressed(list(points))


def point_projective_from_bytes(b: bytes) -> G1Point:  # util.py:35-36
    return G1Point.from_compressed_bytes_unchecked(b)


def field_to_bytes(field: Scalar) -> bytes:  # util.py:39-40
    return bytes(field.to_le_bytes())


def get_random_point() -> G1Point:  # util.py:67-68
    return G1 * random_scalar()
