"""SURVEY 8(f) row 3 against the oracle: the statement rows of the shuffle verifier -- the scalar every wire point and every
CRS point carries once the reference's eight accumulate_check calls and four same-scalar equalities are weighted and summed --
from oracle/shuffle_rows.py (a big-integer restatement of util.py:71-78, ipa.py:164-236, same_msm.py:155-227,
grand_prod.py:137-166, same_perm.py:91-107, same_scalar.py:82-108, curdleproofs.py:176-243) fed with the challenges the
REFERENCE verifier drew (recorded in tests/golden/shuffle_vectors.json), compared byte for byte with

  * the product's host front-end (cg1_shuffle_prepare), on the CPU, and
  * the device row builder k_shuffle_rows (cg1_shuffle_prepare_inputs + cg1_shuffle_rows_device), on the GPU,

for every golden case (ell = 4, 12, 28, 60, 124, 124)."""
import ctypes
import json
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")) as f:
        return json.load(f)


def oracle_rows(case, weights: bytes):
    from oracle import shuffle_rows as SR

    ell = case["ell"]
    rho = [int.from_bytes(weights[32 * k: 32 * k + 32], "little") for k in range(12)]
    ch = [(lab, int.from_bytes(bytes.fromhex(v), "little")) for lab, v in case["challenges"]]
    own, crs, aux = SR.statement_rows(ell, SR.proof_fields(bytes.fromhex(case["proof"]), ell), ch, rho)
    enc = lambda row: b"".join(s.to_bytes(32, "little") for s in row)
    return enc(own), enc(crs), aux


def test_verification_scalars_restatement_is_self_consistent():
    """util.py:71-78 / ipa.py:179-186: s_i is the product of the challenges at the set bits of i (MSB first); s_inv its inverse."""
    from oracle import shuffle_rows as SR

    rng = random.Random(1)
    gam = [rng.randrange(1, SR.R) for _ in range(5)]
    s = SR.vec_s_from(gam, 32)
    assert s[0] == 1 and s[1] == gam[4] and s[16] == gam[0] and s[31] == gam[0] * gam[1] % SR.R * gam[2] % SR.R * gam[3] % SR.R * gam[4] % SR.R
    assert SR.verification_scalars_bitstring(8, 3)[5] == [0, 2]


def test_host_front_end_rows_equal_the_oracle_rows(native_lib, gold):
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
    from test_shuffle_verifier import apply_edits

    for case in gold["cases"]:
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), threads=1)
        inst, proofs, _ = v.pack([apply_edits(case, [])])
        for seed in (case["seed"], 99):
            w = v.draw_weights(1, random.Random(seed))
            prep = v.prepare(inst, proofs, 1, weights=w, want_challenges=True)
            assert int(prep.status[0]) == 0
            own, crs, aux = oracle_rows(case, w)
            L, C = v.crs.points_per_proof, v.crs.ncrs
            assert prep.scalars32.raw[: L * 32] == own, case["ell"]
            assert prep.crs_scalars32.raw[: C * 32] == crs, case["ell"]
        v.close()


@pytest.mark.gpu
def test_device_rows_equal_the_oracle_rows(native_lib, gold):
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
    from test_shuffle_rows_gpu import device_rows
    from test_shuffle_verifier import apply_edits

    N = native_lib
    ctx = N.Context(0)
    for case in gold["cases"]:
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]))
        n = 3                                                       # the same proof under three different weight sets
        inst, proofs, _ = v.pack([apply_edits(case, [])] * n)
        w = v.draw_weights(n, random.Random(case["seed"] + 5))
        sc, crs_sum, rows, st, host_st = device_rows(N, ctx, v, inst, proofs, n, w)
        assert st == host_st == [0] * n
        L, C = v.crs.points_per_proof, v.crs.ncrs
        total = [0] * C
        for i in range(n):
            own, crs, _ = oracle_rows(case, w[i * 12 * 32: (i + 1) * 12 * 32])
            assert sc[i * L * 32: (i + 1) * L * 32] == own, (case["ell"], i)
            assert rows[i * C * 32: (i + 1) * C * 32] == crs, (case["ell"], i)
            for k in range(C):
                total[k] += int.from_bytes(crs[32 * k: 32 * k + 32], "little")
        from oracle.shuffle_rows import R
        assert crs_sum == b"".join((t % R).to_bytes(32, "little") for t in total)       # k_crs_row_sum
        v.close()
