"""Real protocol data through the hot path: the final MSMs of the reference's MSMAccumulator.verify()
(msm_accumulator.py:60-68) recorded from the reference's own seeded N=64 / N=128 shuffle proofs and from
malleated proofs that only the accumulator catches (tests/golden/gen_accumulator_golden.py).

CPU part: the oracle re-derives every expected outcome.  GPU part: the HIP MSM (single and batched) must give
the same group element, i.e. accept exactly the valid proofs."""
import ctypes
import json
import os

import pytest

from conftest import raw96
from oracle import bls12_381 as O
from oracle import c_oracle as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def records():
    recs = json.load(open(os.path.join(ROOT, "tests", "golden", "accumulator_vectors.json")))["records"]
    assert sorted((len(r["bases"]), r["accepts"]) for r in recs) == [(307, False), (307, False), (307, True), (627, False), (627, True)]
    return recs


def _inputs(rec):
    pts = [O.g1_decompress(bytes.fromhex(h)) for h in rec["bases"]]
    return b"".join(raw96(p) for p in pts), b"".join(bytes.fromhex(h) for h in rec["scalars"]), len(pts)


def test_oracle_rederives_expected_outcomes(records):
    for rec in records:
        # 5*ell + 7 unique bases: 307 at N=64, 627 at N=128
        assert len(rec["bases"]) == 5 * (rec["N"] - 4) + 7
        p96, s32, n = _inputs(rec)
        got = C.compress(C.compute_msm(p96, s32, n)).hex()
        assert (got == rec["A_c"]) == rec["accepts"], (rec["N"], rec["tamper"])


@pytest.mark.gpu
def test_gpu_msm_on_real_accumulator_data(native_lib, records):
    N = native_lib
    ctx = N.Context(0)
    try:
        outs = []
        for rec in records:
            p96, s32, n = _inputs(rec)
            want = C.compress(C.compute_msm(p96, s32, n))
            blob = ctx.msm_host(p96, s32, n)
            out = ctypes.create_string_buffer(48)
            N.cg1_compress(out, blob)
            assert out.raw == want
            assert (out.raw.hex() == rec["A_c"]) == rec["accepts"]
            outs.append(want)
        # all five as one regime-B batch
        ps, ss, offs = [], [], [0]
        for rec in records:
            p96, s32, n = _inputs(rec)
            ps.append(p96); ss.append(s32); offs.append(offs[-1] + n)
        blobs = ctx.msm_batched_host(b"".join(ps), b"".join(ss), offs)
        for b, want in zip(blobs, outs):
            out = ctypes.create_string_buffer(48)
            N.cg1_compress(out, b)
            assert out.raw == want
    finally:
        ctx.close()


# ---------------------------------------------------------------- accumulate_check call sequences (msm_accumulator.py:37-58)
@pytest.fixture(scope="module")
def sequences():
    seqs = json.load(open(os.path.join(ROOT, "tests", "golden", "accumulator_vectors.json")))["sequences"]
    assert [(q["N"], len(q["calls"]), q["final"]["accepts"]) for q in seqs] == [(128, 8, True), (64, 8, False)]
    return seqs


class _Replay:
    """Stands in for `random`: hands out the recorded random factors, one per accumulate_check (msm_accumulator.py:43)."""

    def __init__(self, calls):
        self.rhos = [int.from_bytes(bytes.fromhex(c["rho"]), "little") for c in calls]

    def randint(self, lo, hi):
        r = self.rhos.pop(0)
        assert lo <= r <= hi
        return r


def test_oracle_accumulator_replays_reference_call_sequence(sequences):
    """oracle.MSMAccumulator fed the reference's own 8 calls (C, bases incl. identities, scalars, rho) ends in exactly the
    state the reference's accumulator ended in: same keys in the same order, same merged scalars, same A_c, same verdict."""
    for q in sequences:
        acc = O.MSMAccumulator(rng=_Replay(q["calls"]))
        npairs = nident = 0
        for c in q["calls"]:
            bases = [O.g1_decompress(bytes.fromhex(h)) for h in c["bases"]]
            npairs += len(bases)
            nident += sum(1 for b in bases if b is None)
            acc.accumulate_check(O.g1_decompress(bytes.fromhex(c["C"])), bases, [int.from_bytes(bytes.fromhex(h), "little") for h in c["scalars"]])
        ell = q["N"] - 4
        assert npairs == 3 * ell + 5 * q["N"] + 1 and nident == 6        # 1 013 pairs at N=128, 6 of them Z1
        f = q["final"]
        assert [k.hex() for k in acc.base_scalar_map.keys()] == f["bases"]
        assert [v.to_bytes(32, "little").hex() for v in acc.base_scalar_map.values()] == f["scalars"]
        assert O.g1_compress(O.jac_to_affine(acc.A_c)).hex() == f["A_c"]
        pts = b"".join(raw96(O.g1_decompress(k)) for k in acc.base_scalar_map.keys())
        sc = b"".join(v.to_bytes(32, "little") for v in acc.base_scalar_map.values())
        assert (C.compress(C.msm_bucket(pts, sc, len(acc.base_scalar_map))).hex() == f["A_c"]) == f["accepts"]


@pytest.mark.gpu
def test_product_accumulator_replays_reference_call_sequence_on_gpu(native_lib, sequences, monkeypatch):
    """The PRODUCT's MSMAccumulator (Python face -> C ABI -> HIP MSM) fed the same recorded calls: same merged map, same
    A_c, verify() raises exactly when the reference's did."""
    import curdleproofs_pie_amd.msm_accumulator as A
    from curdleproofs_pie_amd.py_arkworks_bls12381 import G1Point, Scalar

    for q in sequences:
        rhos = [Scalar.from_le_bytes(bytes.fromhex(c["rho"])) for c in q["calls"]]
        monkeypatch.setattr(A, "random_scalar", lambda rhos=rhos: rhos.pop(0))
        acc = A.MSMAccumulator()
        for c in q["calls"]:
            acc.accumulate_check(G1Point.from_compressed_bytes_unchecked(bytes.fromhex(c["C"])),
                                 [G1Point.from_compressed_bytes_unchecked(bytes.fromhex(h)) for h in c["bases"]],
                                 [Scalar.from_le_bytes(bytes.fromhex(h)) for h in c["scalars"]])
        assert not rhos                                               # exactly one draw per call
        f = q["final"]
        assert [k.hex() for k in acc.base_scalar_map.keys()] == f["bases"]
        assert [e[0].to_bytes(32, "little").hex() for e in acc.base_scalar_map.values()] == f["scalars"]
        assert bytes(acc.A_c.to_compressed_bytes()).hex() == f["A_c"]
        if f["accepts"]:
            acc.verify()
        else:
            with pytest.raises(AssertionError):
                acc.verify()
