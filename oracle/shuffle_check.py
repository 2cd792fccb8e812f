"""TEST INFRASTRUCTURE (CPU): evaluate the MSM statement a shuffle front-end emits with the CPU oracle.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product evaluates the
statement on the GPU (curdleproofs_pie_amd/shuffle_verifier.py).  A proof is accepted by the reference
(whisk_interface.py:72-109) iff  sum own_scalars*own_points + sum crs_scalars*crs_points  is the identity.

Nothing here imports the product: wire points are decoded by the oracle's own decoder (oracle/msm_oracle.c
orc_decompress == oracle/bls12_381.py g1_decompress, unchecked as util.py:35-36) and the sum is the oracle's bucket MSM.
The caller passes plain byte buffers.
"""
from . import c_oracle


def decompress_affine(data48: bytes, n: int):
    """-> (affine96 bytes, per-point ok flags), oracle decoder, no subgroup test (util.py:35-36)."""
    out, ok = [], []
    for i in range(n):
        rc, aff = c_oracle.decompress(data48[48 * i: 48 * i + 48], False)
        ok.append(rc == 0)
        out.append(aff)
    return b"".join(out), ok


def statement_verdicts(crs48: bytes, n: int, points_per_proof: int, points48: bytes, scalars32: bytes, crs_scalars32: bytes, status):
    """Per-proof verdicts of n prepared statements.  crs48: the `ncrs` CRS points in wire form; points48 / scalars32:
    n x points_per_proof own points (wire form) and their scalars; crs_scalars32: n x ncrs; status: n front-end codes
    (non-zero = already rejected)."""
    L = points_per_proof
    C = len(crs48) // 48
    crs96, crs_ok = decompress_affine(crs48, C)
    assert all(crs_ok), "invalid CRS point"
    res = []
    for i in range(n):
        if status[i]:
            res.append(False)
            continue
        pts, ok = decompress_affine(points48[i * L * 48: (i + 1) * L * 48], L)
        if not all(ok):
            res.append(False)
            continue
        scalars = scalars32[i * L * 32: (i + 1) * L * 32] + crs_scalars32[i * C * 32: (i + 1) * C * 32]
        res.append(c_oracle.msm_bucket(pts + crs96, scalars, L + C) == bytes(96))
    return res


def oracle_verdicts(verifier, prep):
    """Convenience for tests: `verifier` / `prep` are the product's ShuffleBatchVerifier / Prepared objects, read as
    plain data (attribute access only; no product code runs here)."""
    crs = verifier.crs
    return statement_verdicts(crs.bytes[: 48 * crs.ncrs], prep.n, crs.points_per_proof, bytes(prep.points48), bytes(prep.scalars32),
                              bytes(prep.crs_scalars32), [int(prep.status[i]) for i in range(prep.n)])
