"""Drop-in for the third-party wheel `py_arkworks_bls12381` (0.3.5) as curdleproofs.pie uses it.

    from curdleproofs_pie_amd.py_arkworks_bls12381 import G1Point, Scalar

Surface and semantics follow what the reference pins:
  * method set       -- curdleproofs/curdleproofs/test_curdleproofs.py:45-128 (exact `dir()` lists)
  * G1Point behaviour -- test_curdleproofs.py:132-191, stub py_arkworks_bls12381-stubs/__init__.pyi:5-30
  * Scalar behaviour  -- test_curdleproofs.py:194-213, stub :32-54

Single-element operators run in the host C++ of libcurdle_g1.so (a kernel launch per `P + Q` would be
absurd); `G1Point.multiexp_unchecked` and everything batched run on the GPU (msm_accumulator.py).
Values are immutable; every operator returns a new object.

Marshalling (lists of objects -> contiguous buffers) goes through the optional C helper `_pyface` (csrc/pyface.c): one call per
list, straight into page-locked staging.  It moves bytes only; without it the same functions run as Python loops.
"""
from __future__ import annotations

import ctypes
from typing import Any, Iterable, List

from . import _native as N

try:
    from . import _pyface
except ImportError:  # not built (no compiler / Python.h): pure-Python packing below
    _pyface = None

CURVE_ORDER = 52435875175126190479447740508185965837690552500527637822603658699938581184513  # util.py:7

_G1_DIR = [  # test_curdleproofs.py:47-84
    "__add__", "__class__", "__delattr__", "__dir__", "__doc__", "__eq__", "__format__", "__ge__",
    "__getattribute__", "__gt__", "__hash__", "__init__", "__init_subclass__", "__le__", "__lt__", "__module__",
    "__mul__", "__ne__", "__neg__", "__new__", "__radd__", "__reduce__", "__reduce_ex__", "__repr__", "__rmul__",
    "__rsub__", "__setattr__", "__sizeof__", "__str__", "__sub__", "__subclasshook__", "from_compressed_bytes",
    "from_compressed_bytes_unchecked", "identity", "multiexp_unchecked", "to_compressed_bytes",
]
_SCALAR_DIR = [  # test_curdleproofs.py:87-128
    "__add__", "__class__", "__delattr__", "__dir__", "__doc__", "__eq__", "__format__", "__ge__",
    "__getattribute__", "__gt__", "__hash__", "__init__", "__init_subclass__", "__int__", "__le__", "__lt__",
    "__module__", "__mul__", "__ne__", "__neg__", "__new__", "__radd__", "__reduce__", "__reduce_ex__", "__repr__",
    "__rmul__", "__rsub__", "__rtruediv__", "__setattr__", "__sizeof__", "__str__", "__sub__", "__subclasshook__",
    "__truediv__", "from_le_bytes", "inverse", "is_zero", "pow", "square", "to_le_bytes",
]


class _PinnedDir(type):
    """`dir(cls)` reports exactly the wheel's public surface (the reference snapshots it)."""

    def __dir__(cls):
        return list(cls._PINNED_DIR)


def _new_blob():
    return ctypes.create_string_buffer(N.POINT_BYTES)


_set = object.__setattr__


class Scalar(metaclass=_PinnedDir):
    """Element of Fr (integers mod r).  `Scalar(int)` reduces mod r for any size (test_curdleproofs.py:201-207)."""

    _PINNED_DIR = _SCALAR_DIR
    __slots__ = ("_v",)

    def __init__(self, value: int = 0) -> None:
        if isinstance(value, Scalar):
            value = value._v
        if not isinstance(value, int):
            raise TypeError("Scalar() argument must be an int")
        if value < 0:
            raise OverflowError("can't convert negative int to unsigned")
        object.__setattr__(self, "_v", value % CURVE_ORDER)

    def __setattr__(self, k, v):
        raise AttributeError("Scalar is immutable")

    @staticmethod
    def _raw(v: int) -> "Scalar":
        s = object.__new__(Scalar)
        object.__setattr__(s, "_v", v)
        return s

    @staticmethod
    def _coerce(o: Any) -> int:
        if isinstance(o, Scalar):
            return o._v
        raise TypeError(f"unsupported operand type for Scalar arithmetic: {type(o).__name__}")

    def __add__(self, o): return Scalar._raw((self._v + Scalar._coerce(o)) % CURVE_ORDER)
    def __radd__(self, o): return Scalar._raw((Scalar._coerce(o) + self._v) % CURVE_ORDER)
    def __sub__(self, o): return Scalar._raw((self._v - Scalar._coerce(o)) % CURVE_ORDER)
    def __rsub__(self, o): return Scalar._raw((Scalar._coerce(o) - self._v) % CURVE_ORDER)
    def __neg__(self): return Scalar._raw((-self._v) % CURVE_ORDER)

    def __mul__(self, o):
        if isinstance(o, G1Point):
            return o.__mul__(self)
        return Scalar._raw(self._v * Scalar._coerce(o) % CURVE_ORDER)

    def __rmul__(self, o): return self.__mul__(o)
    def __truediv__(self, o): return Scalar._raw(self._v * pow(Scalar._coerce(o), -1, CURVE_ORDER) % CURVE_ORDER) if Scalar._coerce(o) else _div_by_zero()
    def __rtruediv__(self, o): return Scalar._raw(Scalar._coerce(o) * pow(self._v, -1, CURVE_ORDER) % CURVE_ORDER) if self._v else _div_by_zero()
    def __eq__(self, o): return isinstance(o, Scalar) and o._v == self._v
    def __ne__(self, o): return not self.__eq__(o)
    def __hash__(self): return hash(("Fr", self._v))
    def __int__(self): return self._v
    def __str__(self): return self.to_le_bytes().hex()
    def __repr__(self): return f"Scalar({self._v})"

    def inverse(self) -> "Scalar":
        # util.py:51-54 wraps this with `assert res * f == Scalar(1)  # fail in case f == 0`
        return Scalar._raw(pow(self._v, -1, CURVE_ORDER)) if self._v else Scalar._raw(0)

    def is_zero(self) -> bool: return self._v == 0
    def square(self) -> "Scalar": return Scalar._raw(self._v * self._v % CURVE_ORDER)

    def pow(self, exp) -> "Scalar":
        e = int(exp) if not isinstance(exp, (list, tuple)) else sum(int(w) << (64 * i) for i, w in enumerate(exp))
        return Scalar._raw(pow(self._v, e, CURVE_ORDER))

    @staticmethod
    def from_le_bytes(data) -> "Scalar":
        data = bytes(data)
        if len(data) != 32:
            raise ValueError("Err From Rust: serialised data seems to be invalid (need 32 bytes)")
        v = int.from_bytes(data, "little")
        if v >= CURVE_ORDER:  # test_curdleproofs.py:210-213
            raise ValueError("Err From Rust: serialised data seems to be invalid")
        return Scalar._raw(v)

    def to_le_bytes(self) -> bytes:
        return self._v.to_bytes(32, "little")


def _div_by_zero():
    raise ZeroDivisionError("division by zero in Fr")


class G1Point(metaclass=_PinnedDir):
    """Element of the BLS12-381 G1 group.  `G1Point()` is the generator (util.py:9)."""

    _PINNED_DIR = _G1_DIR
    # _b: the 144-byte point blob (host Jacobian).  _a / _k: the point's affine96 record and 48-byte compression once some call
    # has normalised it (None until then) -- values are immutable, so CRS points met by every accumulate_check
    # (msm_accumulator.py:54) and every transcript append are normalised once, not once per call.
    __slots__ = ("_b", "_a", "_k")
    __hash__ = None  # unhashable, test_curdleproofs.py:186-188

    def __init__(self) -> None:
        b = _new_blob()
        N.cg1_generator(b)
        _set(self, "_b", b.raw)
        _set(self, "_a", None)
        _set(self, "_k", None)

    def __setattr__(self, k, v):
        raise AttributeError("G1Point is immutable")

    @staticmethod
    def _from_blob(raw: bytes) -> "G1Point":
        p = object.__new__(G1Point)
        _set(p, "_b", raw)
        _set(p, "_a", None)
        _set(p, "_k", None)
        return p

    @staticmethod
    def identity() -> "G1Point":
        b = _new_blob()
        N.cg1_identity(b)
        return G1Point._from_blob(b.raw)

    def __add__(self, o):
        if not isinstance(o, G1Point):
            return NotImplemented
        b = _new_blob()
        N.cg1_add(b, self._b, o._b)
        return G1Point._from_blob(b.raw)

    __radd__ = __add__

    def __sub__(self, o):
        if not isinstance(o, G1Point):
            return NotImplemented
        b = _new_blob()
        N.cg1_sub(b, self._b, o._b)
        return G1Point._from_blob(b.raw)

    def __rsub__(self, o):
        if not isinstance(o, G1Point):
            return NotImplemented
        return o.__sub__(self)

    def __neg__(self):
        b = _new_blob()
        N.cg1_neg(b, self._b)
        return G1Point._from_blob(b.raw)

    def __mul__(self, s):
        if not isinstance(s, Scalar):
            return NotImplemented
        b = _new_blob()
        N.cg1_mul(b, self._b, s._v.to_bytes(32, "little"))
        return G1Point._from_blob(b.raw)

    __rmul__ = __mul__

    def __eq__(self, o):
        return isinstance(o, G1Point) and bool(N.cg1_eq(self._b, o._b))

    def __ne__(self, o):
        return not self.__eq__(o)

    def to_compressed_bytes(self) -> bytes:
        k = self._k
        if k is None:
            out = ctypes.create_string_buffer(48)
            N.cg1_compress(out, self._b)
            k = out.raw
            _set(self, "_k", k)
        return k

    def __str__(self) -> str:  # test_curdleproofs.py:179
        return self.to_compressed_bytes().hex()

    def __repr__(self) -> str:
        return f"G1Point({self})"

    @staticmethod
    def _decompress(data, check: bool) -> "G1Point":
        data = bytes(data)
        if len(data) != 48:
            raise ValueError("Err From Rust: serialised data seems to be invalid (need 48 bytes)")
        b = _new_blob()
        rc = N.cg1_decompress(b, data, 1 if check else 0)
        if rc != N.OK:
            raise ValueError(f"Err From Rust: serialised data seems to be invalid (code {rc})")
        p = G1Point._from_blob(b.raw)
        if not data[0] & 0x40:
            # a finite point decodes from exactly one encoding (compression flag, x < p, the sign bit that selected y), so these 48
            # bytes ARE its compression: to_compressed_bytes() of a decoded point (transcript appends, util.py:27-28) costs nothing.
            # (The infinity flag is honoured whatever the other bits say: only then may the input differ from the canonical c0 00...)
            _set(p, "_k", data)
        return p

    @staticmethod
    def from_compressed_bytes(data) -> "G1Point":
        return G1Point._decompress(data, True)

    @staticmethod
    def from_compressed_bytes_unchecked(data) -> "G1Point":
        return G1Point._decompress(data, False)

    @staticmethod
    def multiexp_unchecked(bases: Iterable["G1Point"], scalars: Iterable[Scalar]) -> "G1Point":
        """sum_i scalars[i] * bases[i] on the GPU (declared at __init__.pyi:28; unused by the reference)."""
        from .msm_accumulator import compute_MSM

        return compute_MSM(bases, scalars)


# ---------------------------------------------------------------- marshalling: lists of objects <-> contiguous buffers
if _pyface is not None:
    _pyface.bind(G1Point, Scalar)


def pack_points(points, addr: int, capacity: int):
    """Write the blobs of `points` (list / tuple of G1Point) to addr + 144 i; returns (n, every blob has Z in {0, 1})."""
    if _pyface is not None:
        return _pyface.pack_points(points, addr, capacity)
    n = len(points)
    if n > capacity:
        raise ValueError("staging buffer too small")
    raw = b"".join(p._b for p in points)
    ctypes.memmove(addr, raw, len(raw))
    return n, all(p._b[96:] in (_MONT_ONE, _ZERO48) for p in points)


def pack_affine(points, addr: int, capacity: int) -> int:
    """Write the affine96 records of `points` (normalising, once per object, those not normalised before) to addr + 96 i."""
    ensure_normalised(points)
    if _pyface is not None:
        return _pyface.pack_affine(points, addr, capacity)
    n = len(points)
    if n > capacity:
        raise ValueError("staging buffer too small")
    raw = b"".join([p._a for p in points])
    ctypes.memmove(addr, raw, len(raw))
    return n


def pack_scalars(scalars, addr: int, capacity: int) -> int:
    """Write int(s) of every Scalar (or plain int) of `scalars` as 32 little-endian bytes to addr + 32 i."""
    if _pyface is not None:
        return _pyface.pack_scalars(scalars, addr, capacity)
    n = len(scalars)
    if n > capacity:
        raise ValueError("staging buffer too small")
    raw = b"".join((s._v if isinstance(s, Scalar) else s).to_bytes(32, "little") for s in scalars)
    ctypes.memmove(addr, raw, len(raw))
    return n


def points_from_blobs(raw, n: int) -> List[G1Point]:
    """n G1Point objects over the consecutive 144-byte blobs of `raw`."""
    if _pyface is not None:
        return _pyface.points_from_blobs(raw, n)
    raw = bytes(raw)
    return [G1Point._from_blob(raw[144 * i: 144 * i + 144]) for i in range(n)]


def ident(points):
    """(n, fingerprint of the element identities) of a list / tuple: the key of the resident-vector cache."""
    if _pyface is not None:
        return _pyface.ident(points)
    return len(points), hash(tuple(map(id, points)))


def same_items(a, b) -> bool:
    if _pyface is not None:
        return _pyface.same_items(a, b)
    return len(a) == len(b) and all(x is y for x, y in zip(a, b))


_MONT_ONE = bytes.fromhex("fdff02000000097602000cc40b00f4ebba58c7535798485f455752705358ce776dec56a2971a075c93e480fac35ef615")
_ZERO48 = bytes(48)


def ensure_normalised(points) -> None:
    """Fill the `_a` (affine96) / `_k` (compressed48) caches of every point that lacks them: ONE shared inversion for the lot
    (cg1_batch_normalize).  After this, p._a and p._k are bytes for every p in points."""
    todo = [p for p in points if p._a is None]
    n = len(todo)
    if n == 0:
        return
    blobs = b"".join([p._b for p in todo])
    aff = ctypes.create_string_buffer(96 * n)
    cmp_ = ctypes.create_string_buffer(48 * n)
    N.cg1_batch_normalize(blobs, n, aff, cmp_)
    aff, cmp_ = aff.raw, cmp_.raw
    for i, p in enumerate(todo):
        _set(p, "_a", aff[96 * i: 96 * i + 96])
        _set(p, "_k", cmp_[48 * i: 48 * i + 48])


def points_to_affine96(points: List[G1Point]) -> bytes:
    """n points -> n affine96 records (one inversion for those not normalised before)."""
    ensure_normalised(points)
    return b"".join([p._a for p in points])


def points_to_compressed(points: List[G1Point]) -> List[bytes]:
    ensure_normalised(points)
    return [p._k for p in points]
