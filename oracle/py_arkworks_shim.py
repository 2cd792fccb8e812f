"""CPU ORACLE (test infrastructure, NOT product code) -- `G1Point` / `Scalar` value classes over oracle/bls12_381.py.

Stands in for the third-party wheel `py_arkworks_bls12381` 0.3.5 (curdleproofs/pyproject.toml:10; Rust, not under
/root/reference, not installable here) when the reference package is imported in the build container to GENERATE
golden fixtures (tests/golden/gen_*.py inject this module as `sys.modules["py_arkworks_bls12381"]`), so that no byte of a
fixture comes out of the product's arithmetic.  Pure-Python big integers: ~3 ms per scalar multiplication, ~12 s per
N=128 shuffle proof -- affordable for fixtures, useless for anything else.

Surface and semantics follow what the reference pins (stub curdleproofs/py_arkworks_bls12381-stubs/__init__.pyi:5-54,
curdleproofs/curdleproofs/test_curdleproofs.py:45-213): `G1Point()` is the generator, `identity()`, `+ - neg * ==`,
48-byte ZCash compression, `str()` = hex of it, unhashable; `Scalar(int)` reduces mod r, 32-byte LE, `ValueError` on
non-canonical input.  `G1 * Scalar` is the MSB-first double-and-add of oracle.bls12_381.jac_mul.

Only tests/, tests/golden/gen_*.py, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import oracle/.
"""
from __future__ import annotations

from typing import Any, Iterable

from . import bls12_381 as O

CURVE_ORDER = O.CURVE_ORDER


class Scalar:
    __slots__ = ("_v",)

    def __init__(self, value: int = 0) -> None:
        if isinstance(value, Scalar):
            value = value._v
        if not isinstance(value, int):
            raise TypeError("Scalar() argument must be an int")
        if value < 0:
            raise OverflowError("can't convert negative int to unsigned")
        self._v = value % CURVE_ORDER  # test_curdleproofs.py:201-207

    @staticmethod
    def _c(o: Any) -> int:
        if isinstance(o, Scalar):
            return o._v
        raise TypeError(f"unsupported operand type for Scalar arithmetic: {type(o).__name__}")

    def __add__(self, o): return Scalar((self._v + Scalar._c(o)) % CURVE_ORDER)
    def __radd__(self, o): return Scalar((Scalar._c(o) + self._v) % CURVE_ORDER)
    def __sub__(self, o): return Scalar((self._v - Scalar._c(o)) % CURVE_ORDER)
    def __rsub__(self, o): return Scalar((Scalar._c(o) - self._v) % CURVE_ORDER)
    def __neg__(self): return Scalar((-self._v) % CURVE_ORDER)

    def __mul__(self, o):
        if isinstance(o, G1Point):
            return o.__mul__(self)
        return Scalar(self._v * Scalar._c(o) % CURVE_ORDER)

    def __rmul__(self, o): return self.__mul__(o)

    def __truediv__(self, o):
        d = Scalar._c(o)
        if d == 0:
            raise ZeroDivisionError("division by zero in Fr")
        return Scalar(self._v * pow(d, -1, CURVE_ORDER) % CURVE_ORDER)

    def __rtruediv__(self, o):
        if self._v == 0:
            raise ZeroDivisionError("division by zero in Fr")
        return Scalar(Scalar._c(o) * pow(self._v, -1, CURVE_ORDER) % CURVE_ORDER)

    def __eq__(self, o): return isinstance(o, Scalar) and o._v == self._v
    def __ne__(self, o): return not self.__eq__(o)
    def __hash__(self): return hash(("Fr", self._v))
    def __int__(self): return self._v
    def __str__(self): return self.to_le_bytes().hex()
    def __repr__(self): return f"Scalar({self._v})"

    def inverse(self) -> "Scalar":
        return Scalar(pow(self._v, -1, CURVE_ORDER)) if self._v else Scalar(0)   # util.py:51-54 asserts on the caller's side

    def is_zero(self) -> bool: return self._v == 0
    def square(self) -> "Scalar": return Scalar(self._v * self._v % CURVE_ORDER)

    def pow(self, exp) -> "Scalar":
        e = int(exp) if not isinstance(exp, (list, tuple)) else sum(int(w) << (64 * i) for i, w in enumerate(exp))
        return Scalar(pow(self._v, e, CURVE_ORDER))

    @staticmethod
    def from_le_bytes(data) -> "Scalar":
        return Scalar(O.fr_from_le_bytes(data))          # ValueError when >= r (test_curdleproofs.py:210-213)

    def to_le_bytes(self) -> bytes:
        return O.fr_to_le_bytes(self._v)


class G1Point:
    __slots__ = ("_j",)
    __hash__ = None  # unhashable, test_curdleproofs.py:186-188

    def __init__(self) -> None:
        self._j = O.jac_from_affine(O.G1_GEN)            # util.py:9: G1Point() is the generator

    @staticmethod
    def _of(j) -> "G1Point":
        p = object.__new__(G1Point)
        p._j = j
        return p

    @staticmethod
    def identity() -> "G1Point":
        return G1Point._of(O.JAC_INF)

    def __add__(self, o):
        if not isinstance(o, G1Point):
            return NotImplemented
        return G1Point._of(O.jac_add(self._j, o._j))

    __radd__ = __add__

    def __sub__(self, o):
        if not isinstance(o, G1Point):
            return NotImplemented
        return G1Point._of(O.jac_add(self._j, O.jac_neg(o._j)))

    def __rsub__(self, o):
        if not isinstance(o, G1Point):
            return NotImplemented
        return o.__sub__(self)

    def __neg__(self):
        return G1Point._of(O.jac_neg(self._j))

    def __mul__(self, s):
        if not isinstance(s, Scalar):
            return NotImplemented
        return G1Point._of(O.jac_mul(self._j, s._v))

    __rmul__ = __mul__

    def __eq__(self, o):
        return isinstance(o, G1Point) and O.jac_eq(self._j, o._j)

    def __ne__(self, o):
        return not self.__eq__(o)

    def to_compressed_bytes(self) -> bytes:
        return O.g1_compress(O.jac_to_affine(self._j))

    def __str__(self) -> str:  # test_curdleproofs.py:179
        return self.to_compressed_bytes().hex()

    def __repr__(self) -> str:
        return f"G1Point({self})"

    @staticmethod
    def from_compressed_bytes(data) -> "G1Point":
        return G1Point._of(O.jac_from_affine(O.g1_decompress(bytes(data), check_subgroup=True)))

    @staticmethod
    def from_compressed_bytes_unchecked(data) -> "G1Point":
        return G1Point._of(O.jac_from_affine(O.g1_decompress(bytes(data), check_subgroup=False)))

    @staticmethod
    def multiexp_unchecked(bases: Iterable["G1Point"], scalars: Iterable[Scalar]) -> "G1Point":
        acc = O.JAC_INF
        for b, s in zip(bases, scalars):
            acc = O.jac_add(acc, O.jac_mul(b._j, s._v))
        return G1Point._of(acc)
