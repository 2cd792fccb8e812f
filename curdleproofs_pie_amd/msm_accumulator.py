"""GPU-backed `compute_MSM` / `MSMAccumulator` -- same names, arguments and error behaviour as
/root/reference/curdleproofs/curdleproofs/msm_accumulator.py:6-68, with the hot loop on the MI355X.

    from curdleproofs_pie_amd.msm_accumulator import MSMAccumulator, compute_MSM

There is no CPU fallback: without a GPU (or without libcurdle_g1.so) these raise.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Iterable, List, Tuple

from . import _native as N
from . import py_arkworks_bls12381 as B
from .py_arkworks_bls12381 import (CURVE_ORDER, G1Point, Scalar, ensure_normalised, ident, pack_affine, pack_points, pack_scalars, points_from_blobs,
                                   points_to_affine96, points_to_compressed, same_items)
from .util import random_scalar

# Every entry point below that touches the default context -- its staging buffers, its stream, the resident-vector cache -- runs under the
# Python face's one re-entrant lock (py_arkworks_bls12381._LOCK, the lock deferred evaluation uses too): the wheel's values may be
# used from any thread, so may these; calls are serialised per process, which is what one GPU stream does to them anyway.
_LOCK = B._LOCK


def _locked(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **kw):
        with _LOCK:
            return fn(*a, **kw)

    return wrapper
LAZY_MSM_MAX = 2048                      # compute_MSM of up to this many terms returns a deferred value when deferred evaluation is on (= k_msm_small's reach)

_ZERO96 = bytes(96)
_FP = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def _neg_affine96(rec: bytes) -> bytes:
    """(x, y) -> (x, p - y) on an affine96 record (the identity record stays all-zero; y != 0 on this curve)."""
    if rec == _ZERO96:
        return rec
    return rec[:48] + (_FP - int.from_bytes(rec[48:], "little")).to_bytes(48, "little")


# ---------------------------------------------------------------- staging and the resident-vector cache
# One Staging (page-locked host buffers) per context; a small LRU of device-resident point vectors keyed by the IDENTITIES of a
# list's elements (G1Point objects are immutable): the prover passes crs.vec_G / vec_H / vec_R ... to dozens of compute_MSM calls
# (curdleproofs.py:77,94,95,319; grand_prod.py:54,90; ipa.py:97,98) -- from the second sighting on only the scalars are uploaded.
_stagings: Dict[int, "N.Staging"] = {}
_HOST_NORMALISE_MAX = 1024               # up to here compute_MSM normalises on the host (cached per object); above, on the device
_VEC_CACHE_MAX_ENTRIES = 64
_VEC_CACHE_MAX_POINTS = 1 << 23          # ~1 GiB of prepared records on the device, of 288
_vec_cache: "OrderedDict[int, tuple]" = OrderedDict()      # fingerprint -> (tuple of the point objects, N.Vec)
_vec_seen: "OrderedDict[int, None]" = OrderedDict()        # fingerprints met once (a vector is made resident at its second sighting)
_vec_cache_points = 0
last_path = ""                           # which path served the last compute_MSM: "resident" | "blobs" | "blobs_normalised" (tests, bench)


def _staging(ctx) -> "N.Staging":
    st = _stagings.get(id(ctx))
    if st is None or st.ctx is not ctx:
        st = _stagings[id(ctx)] = N.Staging(ctx)
    return st


def _drop_stale(ctx) -> None:
    """Entries made on a context that is gone (N.close_default_context() without release()): free what can be freed, forget the rest."""
    global _vec_cache_points
    for fp in [fp for fp, (_, v) in _vec_cache.items() if v.ctx is not ctx or not v.handle]:
        _, v = _vec_cache.pop(fp)
        _vec_cache_points -= v.n
        if v.ctx.handle:
            v.free()
    for key in [k for k, st in _stagings.items() if st.ctx is not ctx]:
        del _stagings[key]


@_locked
def clear_vec_cache() -> None:
    global _vec_cache_points
    for _, v in _vec_cache.values():
        v.free()
    _vec_cache.clear()
    _vec_seen.clear()
    _vec_cache_points = 0


@_locked
def release() -> None:
    """Free what this module keeps on the default context: the resident vectors and the page-locked staging (before the context is
    closed: N.close_default_context())."""
    clear_vec_cache()
    for st in _stagings.values():
        for b in (st.pts, st.sc):
            if b is not None:
                b.free()
    _stagings.clear()


N.on_close_default_context(release)      # N.close_default_context() frees this module's device / page-locked memory first


def _resident(ctx, bases, n: int):
    """The device-resident form of `bases[:n]` if this very sequence of objects was met before; None the first time."""
    global _vec_cache_points
    if n < 2:
        return None
    _, fp = ident(bases)
    hit = _vec_cache.get(fp)
    if hit is not None:
        if hit[1].ctx is not ctx or not hit[1].handle:
            _drop_stale(ctx)                 # made on a context that was closed since: evict, and treat this sighting as the first
            hit = None
        elif same_items(bases, hit[0]):
            _vec_cache.move_to_end(fp)
            return hit[1]
        else:
            return None
    if fp not in _vec_seen:
        _vec_seen[fp] = None
        if len(_vec_seen) > 4 * _VEC_CACHE_MAX_ENTRIES:
            _vec_seen.popitem(last=False)
        return None
    if n > _VEC_CACHE_MAX_POINTS:
        return None
    st = _staging(ctx)
    addr = st.points(n)
    _, normalised = pack_points(bases, addr, st.cap_pts)
    vec = ctx.vec(addr, n, bool(normalised))
    del _vec_seen[fp]
    _vec_cache[fp] = (tuple(bases), vec)
    _vec_cache_points += n
    while len(_vec_cache) > _VEC_CACHE_MAX_ENTRIES or _vec_cache_points > _VEC_CACHE_MAX_POINTS:
        _, (_, old) = _vec_cache.popitem(last=False)
        _vec_cache_points -= old.n
        old.free()
    return vec


def compute_MSM(bases: Iterable[G1Point], scalars: Iterable[Scalar]) -> G1Point:
    """sum_i scalars[i] * bases[i]  (msm_accumulator.py:6-12).

    `zip` semantics like the reference: any iterables, truncated to the shorter (the reference's tests pass a
    `map` object, test_curdleproofs.py:432).  The arguments are not mutated; a base list met twice is kept (as a tuple of its
    objects) by the resident-vector cache until it is evicted.

    With deferred evaluation on (the default), a call of the protocol's own sizes (<= LAZY_MSM_MAX terms) returns a deferred value:
    the coefficients of deferred bases -- a prover's folded `G_L[i] + G_R[i] * gamma` (ipa.py:142-146) -- are folded into the scalars,
    and the MSM runs on the GPU, together with the other values of its group (the four L / R points of a halving round: one launch),
    when its bytes are needed.  Larger calls run at once: two C walks over the lists straight into page-locked staging (csrc/pyface.c),
    no field arithmetic on the host -- the point blobs go up as the objects hold them and are normalised on the device
    (k_prepare_blobs); a base list met before (same objects) is already resident there and only the scalars move.
    """
    global last_path
    if not (isinstance(bases, (list, tuple)) and isinstance(scalars, (list, tuple))):
        pairs = list(zip(bases, scalars))  # generators, map objects, ...
        bases = [p for p, _ in pairs]
        scalars = [s for _, s in pairs]
    n = min(len(bases), len(scalars))
    if n == 0:
        return G1Point.identity()  # msm_accumulator.py:9
    if len(bases) != n:
        bases = bases[:n]
    if len(scalars) != n:
        scalars = scalars[:n]
    with _LOCK:
        ctx = N.default_context()              # no GPU: NativeError here, deferred or not
        if B._LAZY and n <= LAZY_MSM_MAX:
            last_path = "deferred"
            return B.msm_node(bases, scalars, n)
        st = _staging(ctx)
        sc_addr = st.scalars(n)
        if n >= _SLICED_UPLOAD_MIN and B._pyface is not None:
            fp = ident(bases)[1]
            if fp not in _vec_cache and (fp not in _vec_seen or n > _VEC_CACHE_MAX_POINTS):
                _vec_seen[fp] = None             # first sighting of this list of objects (the second makes it resident: _resident)
                if len(_vec_seen) > 4 * _VEC_CACHE_MAX_ENTRIES:
                    _vec_seen.popitem(last=False)
                return _msm_sliced(ctx, st, sc_addr, bases, scalars, n)
        pack_scalars(scalars, sc_addr, st.cap_sc)
        vec = _resident(ctx, bases, n)
        if vec is not None:
            last_path = "resident"
            return G1Point._from_blob(ctx.msm_vec(vec, sc_addr, n))
        pt_addr = st.points(n)
        if n <= _HOST_NORMALISE_MAX:
            # the protocol's own sizes (4 ... 627 terms): the device inversion (one Fermat chain per lane, ~0.45 ms whatever n) costs more
            # than the call; these few points are normalised on the host, once per OBJECT (CRS points come back call after call)
            pack_affine(bases, pt_addr, (st.cap_pts * 3) // 2)
            last_path = "affine"
            return G1Point._from_blob(ctx.msm_affine(pt_addr, sc_addr, n))
        _, normalised = pack_points(bases, pt_addr, st.cap_pts)
        last_path = "blobs_normalised" if normalised else "blobs"
        return G1Point._from_blob(ctx.msm_blobs(pt_addr, sc_addr, n, bool(normalised)))


_SLICED_UPLOAD_MIN = 1 << 15
last_pack_ms = (0.0, 0.0, 0.0)              # (scalars walk, points walk, device call) of the last sliced compute_MSM, for the bench's python_face object


def _msm_sliced(ctx, st, sc_addr: int, bases, scalars, n: int) -> G1Point:
    """compute_MSM over tens of thousands of objects and more: the two lists are gathered SLICE BY SLICE (threaded C walks, csrc/pyface.c)
    into page-locked staging, and each slice's upload (cg1_h2d_async on the context's copy stream) runs while the next slice is gathered;
    the MSM then starts on blobs that are already on the device.  (Taken at the FIRST sighting of a base list; the second makes it
    resident on the device, _resident, and only scalars move from then on.)"""
    global last_path, last_pack_ms
    import time as _t

    step = 1 << 15 if n < (1 << 18) else 1 << 16
    for attempt in (0, 1):
        pt_addr = st.points(n)
        d_pts, d_sc = ctx.stage_reserve(N.POINT_BYTES * n, 32 * n)
        normalised = True
        t_sc = t_pt = 0.0
        try:
            for off in range(0, n, step):
                cnt = min(step, n - off)
                t0 = _t.perf_counter()
                pack_scalars(scalars, sc_addr + 32 * off, cnt, off, cnt)
                t1 = _t.perf_counter()
                ctx.h2d_async(d_sc + 32 * off, sc_addr + 32 * off, 32 * cnt)
                t2 = _t.perf_counter()
                _, nz = pack_points(bases, pt_addr + N.POINT_BYTES * off, cnt, off, cnt, raise_unforced=True)
                t3 = _t.perf_counter()
                ctx.h2d_async(d_pts + N.POINT_BYTES * off, pt_addr + N.POINT_BYTES * off, N.POINT_BYTES * cnt)
                normalised = normalised and bool(nz)
                t_sc += t1 - t0
                t_pt += t3 - t2
            break
        except B.Unforced:
            # a base is still a deferred value: evaluating it uses the very staging the slices are going up through -- let the copies
            # land, evaluate every deferred base (one batch), and gather again from the first slice
            if attempt:
                raise
            ctx.copy_fence()
            ctx.check(N.cg1_stream_sync(ctx.handle))
            B.materialise(bases)
    ctx.copy_fence()
    t4 = _t.perf_counter()
    out = ctx.msm_blobs_device(d_pts, d_sc, n, normalised)
    last_pack_ms = (t_sc * 1e3, t_pt * 1e3, (_t.perf_counter() - t4) * 1e3)
    last_path = "blobs_normalised" if normalised else "blobs"
    return G1Point._from_blob(out)


@_locked
def compute_MSM_batch(jobs: Iterable[Tuple[Iterable[G1Point], Iterable[Scalar]]]) -> List[G1Point]:
    """[compute_MSM(bases, scalars) for (bases, scalars) in jobs] as ONE GPU launch chain (regime B).

    For many independent small MSMs -- e.g. the final MSMs of a batch of proofs' MSMAccumulator.verify()
    (msm_accumulator.py:60-68): BASELINE config 3 is 1024 of them at 627 terms each."""
    all_pts: List[G1Point] = []
    sc_parts: List[bytes] = []
    offsets = [0]
    for bases, scalars in jobs:
        pairs = list(zip(bases, scalars))
        all_pts.extend(p for p, _ in pairs)
        sc_parts.append(_scalars32([s for _, s in pairs]))
        offsets.append(len(all_pts))
    if len(offsets) == 1:
        return []
    if not all_pts:
        return [G1Point.identity() for _ in offsets[1:]]
    blobs = N.default_context().msm_batched_host(points_to_affine96(all_pts), b"".join(sc_parts), offsets)
    return [G1Point._from_blob(b) for b in blobs]


def _scalars32(values) -> bytes:
    """32-byte little-endian encodings of a list of Scalar / int, concatenated (one C walk)."""
    import ctypes

    n = len(values)
    buf = ctypes.create_string_buffer(32 * max(n, 1))
    pack_scalars(values, ctypes.addressof(buf), n)
    return buf.raw[: 32 * n]


def _blobs_from_affine96(raw: bytes, n: int) -> List[G1Point]:
    import ctypes

    blobs = ctypes.create_string_buffer(N.POINT_BYTES * max(n, 1))
    rc = N.cg1_batch_from_affine96(blobs, raw, n)
    assert rc == N.OK
    return points_from_blobs(blobs, n)


@_locked
def batch_mul(bases: Iterable[G1Point], scalars: Iterable[Scalar]) -> List[G1Point]:
    """[b * s for b, s in zip(bases, scalars)] on the GPU (e.g. G_i * beta^-i, grand_prod.py:64-71)."""
    pairs = list(zip(bases, scalars))
    n = len(pairs)
    if n == 0:
        return []
    raw = N.default_context().batch_mul_add_host(points_to_affine96([p for p, _ in pairs]), n,
                                                 _scalars32([s for _, s in pairs]), n, None, n)
    return _blobs_from_affine96(raw, n)


@_locked
def batch_mul_same_scalar(bases: Iterable[G1Point], scalar: Scalar) -> List[G1Point]:
    """[b * scalar for b in bases] on the GPU (e.g. vec_T = [R * k ...], curdleproofs.py:310-311)."""
    bases = list(bases)
    n = len(bases)
    if n == 0:
        return []
    raw = N.default_context().batch_mul_add_host(points_to_affine96(bases), n, scalar._v.to_bytes(32, "little"), 1, None, n)
    return _blobs_from_affine96(raw, n)


@_locked
def batch_fold(left: Iterable[G1Point], right: Iterable[G1Point], scalar: Scalar) -> List[G1Point]:
    """[l + r * scalar for l, r in zip(left, right)] on the GPU (the IPA / same-MSM folding step,
    ipa.py:142-146, same_msm.py:122-126)."""
    pairs = list(zip(left, right))
    n = len(pairs)
    if n == 0:
        return []
    raw = N.default_context().batch_mul_add_host(points_to_affine96([r for _, r in pairs]), n, scalar._v.to_bytes(32, "little"), 1,
                                                 points_to_affine96([l for l, _ in pairs]), n)
    return _blobs_from_affine96(raw, n)


@_locked
def batch_fold_scalars(left: Iterable[G1Point], right: Iterable[G1Point], scalars: Iterable[Scalar]) -> List[G1Point]:
    """[l + r * s for l, r, s in zip(left, right, scalars)] as one launch: the folds of several provers' rounds (each with its
    own challenge) side by side."""
    trip = list(zip(left, right, scalars))
    n = len(trip)
    if n == 0:
        return []
    raw = N.default_context().batch_mul_add_host(points_to_affine96([r for _, r, _ in trip]), n, _scalars32([s for _, _, s in trip]), n,
                                                 points_to_affine96([l for l, _, _ in trip]), n)
    return _blobs_from_affine96(raw, n)


@_locked
def batch_sum(groups: Iterable[Iterable[G1Point]]) -> List[G1Point]:
    """[reduce(lambda a, b: a + b, g, Z1) for g in groups] on the GPU -- the linear point sums G_sum / H_sum of the CRS
    (crs.py:64-65: over vec_G and vec_H), one wave per group (k_batch_sum)."""
    groups = [list(g) for g in groups]
    if not groups:
        return []
    offsets = [0]
    for g in groups:
        offsets.append(offsets[-1] + len(g))
    flat = [p for g in groups for p in g]
    raw = N.default_context().batch_sum_host(points_to_affine96(flat) if flat else b"", offsets)
    return _blobs_from_affine96(raw, len(groups))


@_locked
def batch_from_compressed(encodings: Iterable[bytes], checked: bool = False) -> List[G1Point]:
    """[G1Point.from_compressed_bytes[_unchecked](e) for e in encodings] with the square roots (and subgroup
    checks) on the GPU -- e.g. the 4*ell tracker points + every proof element of IsValidWhiskShuffleProof
    (whisk_interface.py:96-106, util.py:143-147).  Raises ValueError on the first invalid encoding."""
    enc = [bytes(e) for e in encodings]
    if any(len(e) != 48 for e in enc):
        raise ValueError("Err From Rust: serialised data seems to be invalid (need 48 bytes)")
    if not enc:
        return []
    raw = N.default_context().batch_decompress_host(b"".join(enc), len(enc), checked)
    return _blobs_from_affine96(raw, len(enc))


@_locked
def batch_to_compressed(points: Iterable[G1Point]) -> List[bytes]:
    """[p.to_compressed_bytes() for p in points] (util.py:27-28, points_projective_to_bytes util.py:31-32): ONE batched
    normalisation on the host (a single field inversion) and the encoding itself on the GPU (k_batch_compress)."""
    pts = list(points)
    if not pts:
        return []
    ctx = N.default_context()
    n = len(pts)
    d_in, d_out = ctx.alloc(96 * n), ctx.alloc(48 * n)
    d_in.upload(points_to_affine96(pts))
    ctx.check(N.cg1_batch_compress_device(ctx.handle, d_in.ptr, d_out.ptr, n))
    raw = d_out.download(48 * n)
    return [raw[48 * i: 48 * i + 48] for i in range(n)]


class MSMAccumulator:
    """Random-linear-combination batching of `C == MSM(bases, scalars)` checks (msm_accumulator.py:32-68).

    Same observable behaviour: one `random_scalar()` draw per `accumulate_check` (so seeded runs draw in the
    same order, :43), identity bases skipped (:49-50), equal bases merged by their 48-byte compression (:54-58),
    `verify()` raises AssertionError on mismatch (:68) and ValueError when nothing was accumulated (:63).
    Internals differ where the reference wastes work (its own TODO at :52-53): a call only records its arguments (the points may
    still be deferred values or undecoded encodings); the keys of ALL calls come from ONE evaluation + ONE batched normalisation when
    the map is first needed, the affine form is kept next to the key so nothing is decompressed again (:65),
    and the left-hand sides `rho_i * C_i` join the final GPU MSM instead of costing a scalar-mul each (:45):
        sum_j s_j B_j - sum_i rho_i C_i == 0.
    """

    def __init__(self) -> None:
        self._calls: List[tuple] = []                      # ([bases], [scalar ints], rho) not merged yet
        self._lhs_pts: List[Tuple[G1Point, int]] = []      # (C_i, rho_i), in call order; C_i may still be a deferred value
        self._map: Dict[bytes, List] = {}                  # compressed48 -> [scalar int, affine96]

    def _settle(self) -> None:
        """Merge the recorded calls into the map: every base of every call evaluated / decoded / normalised in one go."""
        if not self._calls:
            return
        with _LOCK:
            calls, self._calls = self._calls, []
            pts = []
            for bases, _, _ in calls:
                pts.extend(bases)
            ensure_normalised(pts)             # ONE decoding + ONE inversion for the points not met before; CRS points keep their normal form
            m = self._map
            for bases, svals, rho in calls:
                for base, sv in zip(bases, svals):
                    a = base._a
                    if a == _ZERO96:  # :49-50 zero bases contribute nothing
                        continue
                    ent = m.get(base._k)  # :54 the 48-byte compression is the key
                    if ent is None:
                        m[base._k] = [rho * sv % CURVE_ORDER, a]
                    else:
                        ent[0] = (ent[0] + rho * sv) % CURVE_ORDER  # :58

    @property
    def base_scalar_map(self) -> Dict[bytes, List]:
        self._settle()
        return self._map

    @property
    def _lhs(self) -> List[Tuple[bytes, int]]:
        """(affine96 of C_i, rho_i) per call (evaluates the C_i that are still deferred)."""
        with _LOCK:
            ensure_normalised([C for C, _ in self._lhs_pts])
        return [(C._a, rho) for C, rho in self._lhs_pts]

    @property
    def A_c(self) -> G1Point:  # the reference's running left-hand side (:45); computed on demand
        lhs = self._lhs
        n = len(lhs)
        if n == 0:
            return G1Point.identity()
        with _LOCK:
            out = N.default_context().msm_host(b"".join(a for a, _ in lhs), _scalars32([r for _, r in lhs]), n)
        return G1Point._from_blob(out)

    def accumulate_check(self, C: G1Point, bases: Iterable[G1Point], scalars: Iterable[Scalar]) -> None:
        random_factor = random_scalar()  # :43  (exactly one draw per call)
        rho = random_factor._v
        if type(C) is not G1Point:
            raise TypeError("accumulate_check: C must be a G1Point")
        bl, sl = [], []
        for b, s in zip(bases, scalars):  # :47
            if type(b) is not G1Point or type(s) is not Scalar:
                raise TypeError("accumulate_check: bases must be G1Point, scalars must be Scalar")
            bl.append(b)
            sl.append(s._v)
        self._lhs_pts.append((C, rho))
        self._calls.append((bl, sl, rho))
        if not B._LAZY:
            self._settle()

    def _final_msm_terms(self) -> Tuple[bytes, bytes, int]:
        """Points and scalars of  sum_j s_j B_j - sum_i rho_i C_i.  A left-hand side that is still a deferred value over bases of G1
        (tested once per base, one pooled call) is not evaluated at all: its terms join the sum with rho_i folded into their
        coefficients; over a base outside G1 it is evaluated first, as the reference computes it (:45)."""
        with _LOCK:
            ents = list(self.base_scalar_map.values())
            pts = [e[1] for e in ents]
            sc = [e[0] for e in ents]
            deferred = [C for C, _ in self._lhs_pts if C._t is not None]
            if deferred:
                B.certify_all(deferred)
                rest = [C for C in deferred if C._sg is not True]
                if rest:
                    B.materialise(rest)
            need = []
            for C, _ in self._lhs_pts:
                t = C._t
                if t is None:
                    need.append(C)
                else:
                    need.extend(t[1])
            ensure_normalised(need)
            R = CURVE_ORDER
            for C, rho in self._lhs_pts:
                t = C._t
                if t is None:
                    # - (rho * C) as rho * (-C): exact for EVERY curve point ((r - rho) * C is -(rho * C) only for C of order r, and the
                    # reference's callers pass points decoded unchecked)
                    pts.append(_neg_affine96(C._a))
                    sc.append(rho)
                else:
                    for c, l in zip(t[0], t[1]):
                        pts.append(l._a)
                        sc.append((-rho * c) % R)
            return b"".join(pts), _scalars32(sc), len(sc)

    @staticmethod
    def verify_many(accumulators: List["MSMAccumulator"]) -> List[bool]:
        """`verify()` of many independent accumulators (one per proof) in ONE batched GPU call.
        Returns one bool per accumulator instead of raising, so a bad proof does not hide the others."""
        for a in accumulators:
            if not a.base_scalar_map:
                raise ValueError("not enough values to unpack (expected 2, got 0)")
        if not accumulators:
            return []
        pts, sc, offsets = [], [], [0]
        for a in accumulators:
            p, s, n = a._final_msm_terms()
            pts.append(p); sc.append(s); offsets.append(offsets[-1] + n)
        with _LOCK:
            blobs = N.default_context().msm_batched_host(b"".join(pts), b"".join(sc), offsets)
        return [N.cg1_is_identity(b) == 1 for b in blobs]

    def verify(self) -> None:
        if not self.base_scalar_map:
            # the reference unpacks `zip(*{}.items())` into two names (:63)
            raise ValueError("not enough values to unpack (expected 2, got 0)")
        pts, sc, n = self._final_msm_terms()
        with _LOCK:
            out = N.default_context().msm_host(pts, sc, n)
        assert N.cg1_is_identity(out) == 1  # computed == self.A_c  (:68)
