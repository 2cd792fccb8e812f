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
import itertools
import os
import threading
import weakref
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


# ---------------------------------------------------------------- deferred evaluation
# The reference's callers use the operators one element at a time -- `G_L[i] + G_R[i] * gamma` in a Python loop (ipa.py:142-146,
# same_msm.py:122-126), `R * k` for every tracker (curdleproofs.py:310-311), 585 single `from_compressed_bytes_unchecked` per verification
# (whisk_interface.py:96-106, util.py:35-36) -- and that control flow is to stay as it is (BASELINE.json north_star).  So the operators do
# not compute: they return DEFERRED values, and a whole batch is evaluated -- by one native call -- when somebody needs bytes or a
# comparison (`to_compressed_bytes`, `==`, `str`, the final MSM of an accumulator).
#
#   leaf            a G1Point that holds its 144-byte blob (`_blob`), or a validated encoding whose y is not computed yet (`_k` set, `_blob`
#                   None: from_compressed_bytes_unchecked checks flags, x < p and the curve equation by a Jacobi symbol, so ValueError is
#                   raised where the wheel raises it; the square roots of all pending points are taken together, later)
#   deferred value  `_t` = (coefs, leaves, from_msm): sum_i coefs[i] * leaves[i] over leaves, integer coefficients |c| < r
#
# `P * s`, `+`, `-`, unary `-` and compute_MSM over deferred bases fold into the coefficient lists (what prover_kernels.py does by hand).
# Coefficients may only be reduced mod r -- `(P * a) * b == P * (a b mod r)` -- for bases of the prime-order subgroup; the reference
# decodes unchecked, so a base's membership (`_sg`: True / False / None = not known yet) is established by one batched endomorphism
# test the first time a product of products needs it; over a base outside G1 the inner value is evaluated first, as the wheel would.
# A/B switch: CURDLE_G1_LAZY=0 or set_lazy(False) -- every operator then computes at once on the host library (round-4 behaviour).
_LOCK = threading.RLock()        # one lock for everything that evaluates or touches the default context's staging (msm_accumulator.py too)
_LAZY = os.environ.get("CURDLE_G1_LAZY", "1") != "0"
_STORE = getattr(_pyface, "store", None) if _pyface is not None else None      # C helper: results of a batched evaluation back into the objects
_DECODE_GPU_MIN = int(os.environ.get("CURDLE_G1_DECODE_GPU_MIN", "192"))      # smallest batch of encodings decoded on the GPU (A/B switch: a huge value = never)
_GLV = os.environ.get("CURDLE_G1_GLV", "1") != "0"      # flushes whose bases are all certified in G1 may use the endomorphism split (A/B switch)
_SIBLING_LOOKBEHIND = 0          # a flush takes the asked-for value and every live deferred value created after it (+ this many before)
_pending: list = []              # weak references to deferred values, in creation order
_next_seq = itertools.count(1).__next__
_ref = weakref.ref
stats = {"flushes": 0, "flushed_values": 0, "flush_terms": 0, "flush_host": 0, "flush_device": 0, "flush_hybrid": 0, "flush_split": 0, "decoded": 0, "decode_batches": 0, "decode_device": 0, "subgroup_tests": 0, "subgroup_device": 0, "subgroup_host": 0}


def set_lazy(on: bool) -> bool:
    """Switch deferred evaluation on / off (A/B); returns the previous setting.  Values already deferred stay valid."""
    global _LAZY
    prev, _LAZY = _LAZY, bool(on)
    return prev


def lazy_enabled() -> bool:
    return _LAZY


class G1Point(metaclass=_PinnedDir):
    """Element of the BLS12-381 G1 group.  `G1Point()` is the generator (util.py:9)."""

    _PINNED_DIR = _G1_DIR
    # _blob: the 144-byte point blob (host Jacobian) or None while the value is deferred.  _a / _k: the point's affine96 record and
    # 48-byte compression once known (None until then) -- values are immutable, so CRS points met by every accumulate_check
    # (msm_accumulator.py:54) and every transcript append are normalised once, not once per call.  _t: the deferred terms (above).
    # _sg: in the prime-order subgroup? (True / False / None = unknown).  _seq: creation number of a deferred value.
    __slots__ = ("_blob", "_a", "_k", "_t", "_sg", "_seq", "__weakref__")
    __hash__ = None  # unhashable, test_curdleproofs.py:186-188

    def __init__(self) -> None:
        b = _new_blob()
        N.cg1_generator(b)
        _set(self, "_blob", b.raw)
        _set(self, "_a", None)
        _set(self, "_k", None)
        _set(self, "_t", None)
        _set(self, "_sg", True)
        _set(self, "_seq", None)

    def __setattr__(self, k, v):
        raise AttributeError("G1Point is immutable")

    @property
    def _b(self) -> bytes:
        """The 144-byte blob; evaluates the value (and the deferred values created after it) if it is still deferred."""
        b = self._blob
        return b if b is not None else _force(self)

    @staticmethod
    def _from_blob(raw: bytes, sg=None) -> "G1Point":
        return _mk(raw, None, None, None, sg, None)

    @staticmethod
    def identity() -> "G1Point":
        return _mk(_IDENTITY_BLOB, None, None, None, True, None)

    def __add__(self, o):
        if type(o) is not G1Point:
            return NotImplemented
        if not _LAZY:
            b = _new_blob()
            N.cg1_add(b, self._b, o._b)
            return _mk(b.raw, None, None, None, True if (self._sg is True and o._sg is True) else None, None)
        ta, tb = self._t, o._t
        sg = True if (self._sg is True and o._sg is True) else None
        if ta is None:
            if tb is None:
                return _mk(None, None, None, ([1, 1], [self, o], False), sg, _next_seq())
            return _mk(None, None, None, ([1] + tb[0], [self] + tb[1], tb[2]), sg, _next_seq())
        if tb is None:
            return _mk(None, None, None, (ta[0] + [1], ta[1] + [o], ta[2]), sg, _next_seq())
        return _mk(None, None, None, (ta[0] + tb[0], ta[1] + tb[1], ta[2] or tb[2]), sg, _next_seq())

    __radd__ = __add__

    def __sub__(self, o):
        if type(o) is not G1Point:
            return NotImplemented
        if not _LAZY:
            b = _new_blob()
            N.cg1_sub(b, self._b, o._b)
            return _mk(b.raw, None, None, None, True if (self._sg is True and o._sg is True) else None, None)
        ta, tb = self._t, o._t
        sg = True if (self._sg is True and o._sg is True) else None
        ca, la, ma = ([1], [self], False) if ta is None else ta
        if tb is None:
            return _mk(None, None, None, (ca + [-1], la + [o], ma), sg, _next_seq())
        return _mk(None, None, None, (ca + [-c for c in tb[0]], la + tb[1], ma or tb[2]), sg, _next_seq())

    def __rsub__(self, o):
        if type(o) is not G1Point:
            return NotImplemented
        return o.__sub__(self)

    def __neg__(self):
        if not _LAZY:
            b = _new_blob()
            N.cg1_neg(b, self._b)
            return _mk(b.raw, None, None, None, self._sg, None)
        t = self._t
        if t is None:
            return _mk(None, None, None, ([-1], [self], False), self._sg, _next_seq())
        return _mk(None, None, None, ([-c for c in t[0]], t[1], t[2]), self._sg, _next_seq())

    def __mul__(self, s):
        if type(s) is not Scalar:
            return NotImplemented
        if not _LAZY:
            b = _new_blob()
            N.cg1_mul(b, self._b, s._v.to_bytes(32, "little"))
            return _mk(b.raw, None, None, None, self._sg, None)
        t = self._t
        if t is None:
            return _mk(None, None, None, ([s._v], [self], False), self._sg, _next_seq())
        if self._sg is not True and not _certify(self):
            # a base outside the prime-order subgroup (or one that cannot be tested): no folding -- the inner value first, as the wheel computes it
            _force(self)
            return _mk(None, None, None, ([s._v], [self], False), self._sg, _next_seq())
        return _mk(None, None, None, (_scale(t[0], s._v, CURVE_ORDER), t[1], t[2]), True, _next_seq())

    __rmul__ = __mul__

    def __eq__(self, o):
        if type(o) is not G1Point:
            return False
        ka, kb = self._k, o._k
        if ka is not None and kb is not None:
            return ka == kb               # both canonical encodings are known (decoded points, normalised values): equal points, equal bytes
        ba, bb = self._blob, o._blob
        if ba is None or bb is None:
            materialise([self, o])
            ba, bb = self._blob, o._blob
        return bool(N.cg1_eq(ba, bb))

    def __ne__(self, o):
        return not self.__eq__(o)

    def to_compressed_bytes(self) -> bytes:
        k = self._k
        if k is None:
            b = self._blob
            if b is None:
                _force(self)              # a flush leaves the normal form and the encoding with every value it evaluates
                k = self._k
                if k is not None:
                    return k
                b = self._blob
            out = ctypes.create_string_buffer(48)
            N.cg1_compress(out, b)
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
        if _LAZY and not check:
            if _pyface is not None:
                return _pyface.decode_lazy(data)
            inf = ctypes.c_int(0)
            rc = N.cg1_validate_compressed(data, ctypes.byref(inf))
            if rc != N.OK:
                raise ValueError(f"Err From Rust: serialised data seems to be invalid (code {rc})")
            if inf.value:
                return _mk(_IDENTITY_BLOB, None, None, None, True, None)
            return _mk(None, None, data, None, None, None)
        b = _new_blob()
        rc = N.cg1_decompress(b, data, 1 if check else 0)
        if rc != N.OK:
            raise ValueError(f"Err From Rust: serialised data seems to be invalid (code {rc})")
        # a finite point decodes from exactly one encoding (compression flag, x < p, the sign bit that selected y), so these 48
        # bytes ARE its compression: to_compressed_bytes() of a decoded point (transcript appends, util.py:27-28) costs nothing.
        # (The infinity flag is honoured whatever the other bits say: only then may the input differ from the canonical c0 00...)
        return _mk(b.raw, None, None if data[0] & 0x40 else data, None, True if (check or data[0] & 0x40) else None, None)

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


def _mk_py(blob, a, k, t, sg, seq) -> G1Point:
    p = object.__new__(G1Point)
    _set(p, "_blob", blob)
    _set(p, "_a", a)
    _set(p, "_k", k)
    _set(p, "_t", t)
    _set(p, "_sg", sg)
    _set(p, "_seq", seq)
    if t is not None:
        _pending.append(_ref(p))
    return p


_ib = _new_blob()
N.cg1_identity(_ib)
_IDENTITY_BLOB = _ib.raw
del _ib

# ---------------------------------------------------------------- marshalling: lists of objects <-> contiguous buffers
def _scale_py(coefs, v, R):
    return [c * v % R for c in coefs]


if _pyface is not None:
    _pyface.bind(G1Point, Scalar, _pending, _IDENTITY_BLOB)
    _pyface.set_native(ctypes.cast(N.lib.cg1_validate_compressed, ctypes.c_void_p).value)
    _mk = _pyface.mk
    _scale = _pyface.scale
    Unforced = _pyface.Unforced
else:
    _mk = _mk_py
    _scale = _scale_py

    class Unforced(LookupError):
        pass


# ---------------------------------------------------------------- evaluation of deferred values
def _have_gpu():
    """The default context's handle, or None on a machine without a GPU (then only operator batches small enough for the host pool --
    and nothing that came out of compute_MSM -- can be evaluated)."""
    if N.cg1_device_count() <= 0:
        return None
    return N.default_context()


def _live_after(seq: int):
    """The live deferred values created after `seq` (and up to _SIBLING_LOOKBEHIND before), newest first; dead tail entries are dropped."""
    out = []
    pend = _pending
    i = len(pend)
    floor = seq - _SIBLING_LOOKBEHIND
    trim = True
    while i > 0:
        nd = pend[i - 1]()
        if nd is None or nd._t is None:
            if trim:
                pend.pop()
            i -= 1
            continue
        trim = False
        if nd._seq < floor:
            break
        out.append(nd)
        i -= 1
    if len(pend) > 8192:          # values that were never asked for (a prover's folded bases) die without being evaluated: compact
        live = []
        for r in pend:
            o = r()
            if o is not None and o._t is not None:
                live.append(r)
        pend[:] = live
    return out


def _force(p: G1Point) -> bytes:
    """Evaluate `p` (a deferred value or an undecoded leaf) -- together with every live deferred value created after it: the reference
    serialises / compares its values in the order it made them, so the first one asked for brings its whole group (the four L / R
    points of a halving round, the 2 x 124 `R * k` of the instance) into ONE native call."""
    with _LOCK:
        b = p._blob
        if b is not None:
            return b
        if p._t is None:
            _decode_leaves([p])
            return p._blob
        group = _live_after(p._seq)
        if not any(g is p for g in group):
            group.append(p)
        _flush(group)
        return p._blob


def materialise(points) -> None:
    """Give every point of `points` its blob: all deferred values among them in one flush, all undecoded leaves in one decoding."""
    with _LOCK:
        nodes, leaves, seen = [], [], set()
        for p in points:
            if p._blob is None and id(p) not in seen:
                seen.add(id(p))
                (leaves if p._t is None else nodes).append(p)
        if nodes:
            first = min(nd._seq for nd in nodes)
            for g in _live_after(first):
                if id(g) not in seen:
                    seen.add(id(g))
                    nodes.append(g)
            _flush(nodes)
        if leaves:
            _decode_leaves(leaves)


def _decode_leaves(leaves) -> None:
    """y for validated encodings: ONE pooled call (csrc/lazy_host.cpp) for the lot; blobs and affine96 records come back."""
    n = len(leaves)
    enc = b"".join([l._k for l in leaves])
    blobs = ctypes.create_string_buffer(N.POINT_BYTES * n)
    aff = ctypes.create_string_buffer(96 * n)
    bad = ctypes.c_size_t(0)
    # 192 .. 8 192 encodings and a GPU: one DPP row per point, the square-root chain with one limb per lane (~0.2 ms whatever n is;
    # the pool needs 0.54 ms for the 585 of one verification); otherwise the host's worker pool
    ctx = _have_gpu() if _DECODE_GPU_MIN <= n <= 8192 else None
    if ctx is not None:
        rc = N.cg1_batch_decompress_rows(ctx.handle, enc, n, blobs, aff, ctypes.byref(bad))
        stats["decode_device"] += 1
    else:
        rc = N.cg1_batch_decompress_pool(enc, n, blobs, aff, 0, ctypes.byref(bad))
    if rc != N.OK:      # cannot happen for encodings that passed the validation
        raise ValueError(f"Err From Rust: serialised data seems to be invalid (point {bad.value}, code {rc})")
    if _STORE is not None:
        _STORE(leaves, blobs, aff, None, False)
    else:
        blobs, aff = blobs.raw, aff.raw
        for i, l in enumerate(leaves):
            _set(l, "_a", aff[96 * i: 96 * i + 96])
            _set(l, "_blob", blobs[144 * i: 144 * i + 144])
    stats["decoded"] += n
    stats["decode_batches"] += 1


def _subgroup_flags(todo) -> None:
    """`_sg` for every point of `todo` (normalised first): one native call -- one wave per point on the GPU for 32 .. 4 096 points
    (k_subgroup_row), the host's worker pool otherwise (or without a GPU)."""
    ensure_normalised(todo)
    n = len(todo)
    flags = ctypes.create_string_buffer(n)
    ctx = _have_gpu()
    used = ctypes.c_int(0)
    rc = N.cg1_batch_subgroup(ctx.handle if ctx is not None else None, b"".join([l._a for l in todo]), n, flags, ctypes.byref(used))
    if rc != N.OK:
        raise N.NativeError(f"cg1_batch_subgroup failed ({rc})")
    for l, f in zip(todo, flags.raw):
        _set(l, "_sg", bool(f))
    stats["subgroup_tests"] += n
    stats["subgroup_device" if used.value else "subgroup_host"] += 1


def _certify(node: G1Point) -> bool:
    """True iff every leaf of the deferred value lies in the prime-order subgroup (then its coefficients may be reduced mod r).  Leaves
    not tested before are tested now, in one pooled call; the verdict stays with the leaf object."""
    with _LOCK:
        leaves = node._t[1] if node._t is not None else [node]
        todo, seen = [], set()
        for l in leaves:
            if l._sg is None and id(l) not in seen:
                seen.add(id(l))
                todo.append(l)
        if todo:
            _subgroup_flags(todo)
        ok = all(l._sg is True for l in leaves)
        if ok:
            _set(node, "_sg", True)
        return ok


def certify_all(points) -> None:
    """Establish `_sg` for every leaf under `points` in one pooled call (a caller that knows it will fold: msm_accumulator.compute_MSM)."""
    with _LOCK:
        todo, seen = [], set()
        for p in points:
            for l in (p._t[1] if p._t is not None else (p,)):
                if l._sg is None and id(l) not in seen:
                    seen.add(id(l))
                    todo.append(l)
        if not todo:
            return
        _subgroup_flags(todo)
        for p in points:
            if p._t is not None and p._sg is not True and all(l._sg is True for l in p._t[1]):
                _set(p, "_sg", True)


def _flush(nodes) -> None:
    """Evaluate deferred values in one native call (cg1_lincomb_batch): the host's worker pool for a handful of operator results, the GPU's
    batched MSM for anything larger and for everything that came out of compute_MSM.  Every value leaves with its blob (Z = 1), its
    affine96 record and its 48-byte compression."""
    if _pyface is not None:
        leaf_list, offs, tba, scb, T, from_msm = _pyface.assemble(nodes, CURVE_ORDER)
        return _flush_run(nodes, leaf_list, offs, tba, scb, T, from_msm)
    index = {}
    leaf_list = []
    offsets = [0]
    tb = []
    sc = []
    from_msm = False
    R = CURVE_ORDER
    for nd in nodes:
        coefs, lv, msm = nd._t
        from_msm = from_msm or msm
        if nd._sg is True:
            acc = {}
            for c, l in zip(coefs, lv):
                i = index.get(id(l))
                if i is None:
                    i = index[id(l)] = len(leaf_list)
                    leaf_list.append(l)
                acc[i] = acc.get(i, 0) + c
            for i, c in acc.items():
                c %= R
                if c:
                    tb.append(i)
                    sc.append(c)
        else:       # some base may lie outside the subgroup: coefficients stay the integers they are (|c| < r), the sign goes to the base
            for c, l in zip(coefs, lv):
                if c == 0:
                    continue
                i = index.get(id(l))
                if i is None:
                    i = index[id(l)] = len(leaf_list)
                    leaf_list.append(l)
                if c < 0:
                    tb.append(i | 0x80000000)
                    sc.append(-c)
                else:
                    tb.append(i)
                    sc.append(c)
        offsets.append(len(tb))
    n_out, T = len(nodes), len(tb)
    offs = (ctypes.c_uint32 * (n_out + 1))(*offsets)
    tba = (ctypes.c_uint32 * max(T, 1))(*tb)
    scb = ctypes.create_string_buffer(32 * max(T, 1))
    pack_scalars(sc, ctypes.addressof(scb), T)
    return _flush_run(nodes, leaf_list, offs, tba, scb, T, from_msm)


def _flush_run(nodes, leaf_list, offs, tba, scb, T: int, from_msm: bool) -> None:
    """The native half of a flush: bases normalised (undecoded leaves decoded in one pooled call), one cg1_lincomb_batch, results stored."""
    ensure_normalised(leaf_list)
    n_out = len(nodes)
    bases = b"".join([l._a for l in leaf_list])
    out_b = ctypes.create_string_buffer(N.POINT_BYTES * n_out)
    out_a = ctypes.create_string_buffer(96 * n_out)
    out_k = ctypes.create_string_buffer(48 * n_out)
    used = ctypes.c_int(0)
    # compute_MSM has no host path: a batch carrying its results needs the GPU context (NativeError without one); the native call then
    # splits the batch by what each engine is good at (csrc/capi_lincomb.h cg1_lincomb_batch: combinations of >= 4 weighted terms on the GPU,
    # the one- to three-term operator results on the host's worker pool meanwhile)
    ctx = N.default_context() if from_msm else _have_gpu()
    handle = ctx.handle if ctx is not None else None
    # every base certified in G1 (generator multiples, results of earlier MSMs, points that passed the subgroup test): the engine may use
    # the endomorphism split for this one call (context parameter "glv", csrc/glv.h); never on a guess -- outside G1 it would be wrong
    split = ctx is not None and _GLV and all([l._sg is True for l in leaf_list])
    if split:
        ctx.set_param("glv", 1)
    try:
        rc = N.cg1_lincomb_batch(handle, bases, len(leaf_list), offs, n_out, tba, scb, 0, out_b, out_a, out_k, ctypes.byref(used))
    finally:
        if split:
            ctx.set_param("glv", 0)
    if split:
        stats["flush_split"] += 1
    if rc != N.OK:
        if ctx is not None:
            ctx.check(rc)
        raise N.NativeError(f"cg1_lincomb_batch failed ({rc})")
    if _STORE is not None:
        _STORE(nodes, out_b, out_a, out_k, True)
    else:
        rb, ra, rk = out_b.raw, out_a.raw, out_k.raw
        for j, nd in enumerate(nodes):
            _set(nd, "_a", ra[96 * j: 96 * j + 96])
            _set(nd, "_k", rk[48 * j: 48 * j + 48])
            _set(nd, "_blob", rb[144 * j: 144 * j + 144])
            _set(nd, "_t", None)
    pend = _pending                  # drop what the tail of the pending list no longer needs: evaluated values, values that died unevaluated
    while pend:
        o = pend[-1]()
        if o is not None and o._t is not None:
            break
        pend.pop()
    stats["flushes"] += 1
    stats["flushed_values"] += n_out
    stats["flush_terms"] += T
    stats[("flush_host", "flush_host", "flush_device", "flush_hybrid")[used.value & 3]] += 1


def msm_node(bases, scalars, n: int) -> G1Point:
    """compute_MSM as a deferred value: sum_i scalars[i] * bases[i] with the coefficients of deferred bases folded into the scalars
    (msm_accumulator.compute_MSM, for the protocol's own sizes).  Deferred bases over points not known to be in G1 are tested (one
    pooled call) and, if a base really lies outside, evaluated first."""
    R = CURVE_ORDER
    if _pyface is not None:
        coefs, leaves, sg, unsure = _pyface.msm_terms(bases, scalars, n, R)
        if unsure is not None:
            certify_all(unsure)
            still = [b for b in unsure if b._sg is not True]
            if still:
                materialise(still)
            coefs, leaves, sg, unsure = _pyface.msm_terms(bases, scalars, n, R)
            assert unsure is None
        return _mk(None, None, None, (coefs, leaves, True), True if sg else None, _next_seq())
    unsure = [b for b in bases if type(b) is G1Point and b._t is not None and b._sg is not True]
    if unsure:
        certify_all(unsure)
        still = [b for b in unsure if b._sg is not True]
        if still:
            materialise(still)
    coefs, leaves = [], []
    sg = True
    for b, s in zip(bases, scalars):
        if type(b) is not G1Point:
            raise TypeError(f"compute_MSM: bases must be G1Point, not {type(b).__name__}")
        if type(s) is Scalar:
            v = s._v
        elif type(s) is int and 0 <= s < R:
            v = s
        else:
            raise TypeError(f"compute_MSM: scalars must be Scalar, not {type(s).__name__}")
        t = b._t
        if t is None:
            coefs.append(v)
            leaves.append(b)
            if b._sg is not True:
                sg = False
        else:
            coefs.extend([c * v % R for c in t[0]])
            leaves.extend(t[1])
    return _mk(None, None, None, (coefs, leaves, True), True if sg else None, _next_seq())


def pack_points(points, addr: int, capacity: int, start: int = 0, count: int = -1, raise_unforced: bool = False):
    """Write the blobs of points[start : start + count] (list / tuple of G1Point; default: all) to addr + 144 i; returns (n, every blob
    has Z in {0, 1}).  Deferred values among them are evaluated first (one batch).  Long ranges are walked by several threads
    (csrc/pyface.c: the objects are immutable and the caller holds the GIL)."""
    if _pyface is not None:
        try:
            return _pyface.pack_points(points, addr, capacity, start, count)
        except Unforced:
            if raise_unforced:              # the caller has uploads in flight that an evaluation would disturb: it evaluates and starts over
                raise
            materialise(points if count < 0 else points[start: start + count])
            return _pyface.pack_points(points, addr, capacity, start, count)
    if count >= 0 or start:
        points = points[start: start + count] if count >= 0 else points[start:]
    n = len(points)
    if n > capacity:
        raise ValueError("staging buffer too small")
    materialise(points)
    raw = b"".join(p._blob for p in points)
    ctypes.memmove(addr, raw, len(raw))
    return n, all(p._blob[96:] in (_MONT_ONE, _ZERO48) for p in points)


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


def pack_scalars(scalars, addr: int, capacity: int, start: int = 0, count: int = -1) -> int:
    """Write int(s) of every Scalar (or plain int) of scalars[start : start + count] (default: all) as 32 little-endian bytes to addr + 32 i."""
    if _pyface is not None:
        return _pyface.pack_scalars(scalars, addr, capacity, start, count)
    if count >= 0 or start:
        scalars = scalars[start: start + count] if count >= 0 else scalars[start:]
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
    """Fill the `_a` (affine96) / `_k` (compressed48) caches of every point that lacks them: deferred values are evaluated (one flush),
    undecoded leaves decoded (one pooled call), the rest normalised with ONE shared inversion (cg1_batch_normalize).  After this, p._a
    and p._k are bytes for every p in points."""
    todo = [p for p in points if p._a is None]
    if not todo:
        return
    with _LOCK:
        if any(p._blob is None for p in todo):
            materialise(todo)
            todo = [p for p in todo if p._a is None]
        n = len(todo)
        if n == 0:
            return
        blobs = b"".join([p._blob for p in todo])
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
