"""GPU-backed `compute_MSM` / `MSMAccumulator` -- same names, arguments and error behaviour as
/root/reference/curdleproofs/curdleproofs/msm_accumulator.py:6-68, with the hot loop on the MI355X.

    from curdleproofs_pie_amd.msm_accumulator import MSMAccumulator, compute_MSM

There is no CPU fallback: without a GPU (or without libcurdle_g1.so) these raise.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Tuple

from . import _native as N
from .py_arkworks_bls12381 import CURVE_ORDER, G1Point, Scalar, points_to_affine96, points_to_compressed
from .util import random_scalar

_ZERO96 = bytes(96)


def compute_MSM(bases: Iterable[G1Point], scalars: Iterable[Scalar]) -> G1Point:
    """sum_i scalars[i] * bases[i]  (msm_accumulator.py:6-12).

    `zip` semantics like the reference: any iterables, truncated to the shorter (the reference's tests pass a
    `map` object, test_curdleproofs.py:432).  Neither argument is retained or mutated.
    """
    pairs = list(zip(bases, scalars))
    n = len(pairs)
    if n == 0:
        return G1Point.identity()  # msm_accumulator.py:9
    pts = points_to_affine96([p for p, _ in pairs])
    sc = b"".join(s._v.to_bytes(32, "little") for _, s in pairs)
    out = N.default_context().msm_host(pts, sc, n)
    return G1Point._from_blob(out)


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
        sc_parts.extend(s._v.to_bytes(32, "little") for _, s in pairs)
        offsets.append(len(all_pts))
    if len(offsets) == 1:
        return []
    if not all_pts:
        return [G1Point.identity() for _ in offsets[1:]]
    blobs = N.default_context().msm_batched_host(points_to_affine96(all_pts), b"".join(sc_parts), offsets)
    return [G1Point._from_blob(b) for b in blobs]


def _blobs_from_affine96(raw: bytes, n: int) -> List[G1Point]:
    import ctypes

    out = []
    for i in range(n):
        b = ctypes.create_string_buffer(N.POINT_BYTES)
        rc = N.cg1_from_affine96(b, raw[96 * i: 96 * i + 96], 0)
        assert rc == N.OK
        out.append(G1Point._from_blob(b.raw))
    return out


def batch_mul(bases: Iterable[G1Point], scalars: Iterable[Scalar]) -> List[G1Point]:
    """[b * s for b, s in zip(bases, scalars)] on the GPU (e.g. G_i * beta^-i, grand_prod.py:64-71)."""
    pairs = list(zip(bases, scalars))
    n = len(pairs)
    if n == 0:
        return []
    raw = N.default_context().batch_mul_add_host(points_to_affine96([p for p, _ in pairs]), n,
                                                 b"".join(s._v.to_bytes(32, "little") for _, s in pairs), n, None, n)
    return _blobs_from_affine96(raw, n)


def batch_mul_same_scalar(bases: Iterable[G1Point], scalar: Scalar) -> List[G1Point]:
    """[b * scalar for b in bases] on the GPU (e.g. vec_T = [R * k ...], curdleproofs.py:310-311)."""
    bases = list(bases)
    n = len(bases)
    if n == 0:
        return []
    raw = N.default_context().batch_mul_add_host(points_to_affine96(bases), n, scalar._v.to_bytes(32, "little"), 1, None, n)
    return _blobs_from_affine96(raw, n)


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


def batch_fold_scalars(left: Iterable[G1Point], right: Iterable[G1Point], scalars: Iterable[Scalar]) -> List[G1Point]:
    """[l + r * s for l, r, s in zip(left, right, scalars)] as one launch: the folds of several provers' rounds (each with its
    own challenge) side by side."""
    trip = list(zip(left, right, scalars))
    n = len(trip)
    if n == 0:
        return []
    raw = N.default_context().batch_mul_add_host(points_to_affine96([r for _, r, _ in trip]), n, b"".join(s._v.to_bytes(32, "little") for _, _, s in trip), n,
                                                 points_to_affine96([l for l, _, _ in trip]), n)
    return _blobs_from_affine96(raw, n)


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
    Internals differ where the reference wastes work (its own TODO at :52-53): keys come from ONE batched
    normalisation per call, the affine form is kept next to the key so nothing is decompressed again (:65),
    and the left-hand sides `rho_i * C_i` join the final GPU MSM instead of costing a scalar-mul each (:45):
        sum_j s_j B_j - sum_i rho_i C_i == 0.
    """

    def __init__(self) -> None:
        self._lhs: List[Tuple[bytes, int]] = []           # (affine96 of C_i, rho_i)
        self.base_scalar_map: Dict[bytes, List] = {}       # compressed48 -> [scalar int, affine96]

    @property
    def A_c(self) -> G1Point:  # the reference's running left-hand side (:45); computed on demand
        n = len(self._lhs)
        if n == 0:
            return G1Point.identity()
        out = N.default_context().msm_host(b"".join(a for a, _ in self._lhs),
                                           b"".join(r.to_bytes(32, "little") for _, r in self._lhs), n)
        return G1Point._from_blob(out)

    def accumulate_check(self, C: G1Point, bases: Iterable[G1Point], scalars: Iterable[Scalar]) -> None:
        random_factor = random_scalar()  # :43  (exactly one draw per call)
        rho = random_factor._v
        pairs = list(zip(bases, scalars))  # :47
        pts = [C] + [b for b, _ in pairs]
        aff = points_to_affine96(pts)
        keys = points_to_compressed(pts)
        self._lhs.append((aff[:96], rho))
        m = self.base_scalar_map
        for i, (_, scalar) in enumerate(pairs, start=1):
            a = aff[96 * i: 96 * i + 96]
            if a == _ZERO96:  # :49-50 zero bases contribute nothing
                continue
            k = keys[i]
            ent = m.get(k)
            if ent is None:
                m[k] = [rho * scalar._v % CURVE_ORDER, a]
            else:
                ent[0] = (ent[0] + rho * scalar._v) % CURVE_ORDER  # :58

    def _final_msm_terms(self) -> Tuple[bytes, bytes, int]:
        ents = list(self.base_scalar_map.values())
        pts = b"".join(e[1] for e in ents) + b"".join(a for a, _ in self._lhs)
        sc = b"".join(e[0].to_bytes(32, "little") for e in ents) + b"".join(
            ((-r) % CURVE_ORDER).to_bytes(32, "little") for _, r in self._lhs)
        return pts, sc, len(ents) + len(self._lhs)

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
        blobs = N.default_context().msm_batched_host(b"".join(pts), b"".join(sc), offsets)
        return [N.cg1_is_identity(b) == 1 for b in blobs]

    def verify(self) -> None:
        if not self.base_scalar_map:
            # the reference unpacks `zip(*{}.items())` into two names (:63)
            raise ValueError("not enough values to unpack (expected 2, got 0)")
        ents = list(self.base_scalar_map.values())
        pts = b"".join(e[1] for e in ents) + b"".join(a for a, _ in self._lhs)
        sc = b"".join(e[0].to_bytes(32, "little") for e in ents) + b"".join(
            ((-r) % CURVE_ORDER).to_bytes(32, "little") for _, r in self._lhs)
        n = len(ents) + len(self._lhs)
        out = N.default_context().msm_host(pts, sc, n)
        assert N.cg1_is_identity(out) == 1  # computed == self.A_c  (:68)
