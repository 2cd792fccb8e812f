"""TEST INFRASTRUCTURE (CPU): evaluate the MSM statement the shuffle front-end emits with the CPU oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product evaluates the
statement on the GPU (curdleproofs_pie_amd/shuffle_verifier.py).  A proof is accepted by the reference
(whisk_interface.py:72-109) iff  sum own_scalars*own_points + sum crs_scalars*crs_points  is the identity.
"""
import ctypes

from . import c_oracle


def host_decompress_affine(data48: bytes, n: int):
    """-> (affine96 bytes, per-point ok flags) with the host codec (cg1_decompress, unchecked: util.py:35-36)."""
    from curdleproofs_pie_amd import _native as N

    out, ok = [], []
    blob = ctypes.create_string_buffer(N.POINT_BYTES)
    aff = ctypes.create_string_buffer(96)
    for i in range(n):
        rc = N.cg1_decompress(blob, data48[48 * i: 48 * i + 48], 0)
        ok.append(rc == 0)
        if rc == 0:
            N.cg1_to_affine96(aff, blob.raw)
            out.append(aff.raw)
        else:
            out.append(bytes(96))
    return b"".join(out), ok


def oracle_verdicts(verifier, prep):
    """Per-proof verdicts of a `Prepared` batch: statement == identity, by the C oracle's bucket MSM."""
    crs = verifier.crs
    L, C = crs.points_per_proof, crs.ncrs
    pts_all, sc_all, cs_all = bytes(prep.points48), bytes(prep.scalars32), bytes(prep.crs_scalars32)
    res = []
    for i in range(prep.n):
        if prep.status[i]:
            res.append(False)
            continue
        pts, ok = host_decompress_affine(pts_all[i * L * 48: (i + 1) * L * 48], L)
        if not all(ok):
            res.append(False)
            continue
        points = pts + crs.affine96
        scalars = sc_all[i * L * 32: (i + 1) * L * 32] + cs_all[i * C * 32: (i + 1) * C * 32]
        res.append(c_oracle.msm_bucket(points, scalars, L + C) == bytes(96))
    return res
