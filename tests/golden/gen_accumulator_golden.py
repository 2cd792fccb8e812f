#!/usr/bin/env python3
"""Record REAL protocol data for the hot path: run the reference's own shuffle prover and verifier
(/root/reference/curdleproofs, imported unmodified in the build container with our G1Point/Scalar module
standing in for the missing Rust wheel) on seeded inputs, and dump what its `MSMAccumulator.verify()`
(msm_accumulator.py:60-68) feeds to `compute_MSM` -- the 5*ell+7 unique bases, the merged scalars -- together
with the left-hand side A_c it is compared with, for valid proofs and for the reference's own tampered cases
(test_curdleproofs.py:643-670).  Only data is written: tests/golden/accumulator_vectors.json
(compressed points 48 B hex, scalars 32 B LE hex).  The GPU tests recompute every MSM and compare with A_c.

The curve arithmetic during generation is our host C++ (the wheel cannot run here), so these are
protocol-shaped regression fixtures whose expected values are additionally re-derived by the CPU oracle in
tests/test_accumulator_golden.py -- not an independent pin of the group law (that is tests/test_oracle_kat.py).

    python tests/golden/gen_accumulator_golden.py
"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/curdleproofs")
sys.path.insert(0, "/root/reference/merlin_transcripts")
import curdleproofs_pie_amd.py_arkworks_bls12381 as backend  # noqa: E402

sys.modules["py_arkworks_bls12381"] = backend

import curdleproofs.msm_accumulator as ref_acc  # noqa: E402
from curdleproofs.crs import CurdleproofsCrs  # noqa: E402
from curdleproofs.curdleproofs import N_BLINDERS, CurdleProofsProof, shuffle_permute_and_commit_input  # noqa: E402
from curdleproofs.util import get_random_point, random_scalar  # noqa: E402

RECORDS = []
_orig_verify = ref_acc.MSMAccumulator.verify


def recording_verify(self):
    keys = list(self.base_scalar_map.keys())
    rec = {
        "bases": [bytes(k).hex() for k in keys],
        "scalars": [bytes(self.base_scalar_map[k].to_le_bytes()).hex() for k in keys],
        "A_c": bytes(self.A_c.to_compressed_bytes()).hex(),
    }
    try:
        _orig_verify(self)
        rec["accepts"] = True
    except AssertionError:
        rec["accepts"] = False
        RECORDS.append(rec)
        raise
    RECORDS.append(rec)


ref_acc.MSMAccumulator.verify = recording_verify


def run(N, seed, tamper):
    random.seed(seed)
    ell = N - N_BLINDERS
    crs = CurdleproofsCrs.new(ell, N_BLINDERS)
    permutation = list(range(ell))
    random.shuffle(permutation)
    k = random_scalar()
    vec_R = [get_random_point() for _ in range(ell)]
    vec_S = [get_random_point() for _ in range(ell)]
    vec_T, vec_U, M, blinders = shuffle_permute_and_commit_input(crs, vec_R, vec_S, permutation, k)
    proof = CurdleProofsProof.new(crs=crs, vec_R=vec_R, vec_S=vec_S, vec_T=vec_T, vec_U=vec_U, M=M,
                                  permutation=permutation, k=k, vec_m_blinders=blinders)
    before = len(RECORDS)
    if tamper == "none":
        proof.verify(crs, vec_R, vec_S, vec_T, vec_U, M)
    elif tamper == "swap_R_S":                       # test_curdleproofs.py:643-644
        try:
            proof.verify(crs, vec_S, vec_R, vec_T, vec_U, M)
            raise SystemExit("tampered proof accepted?!")
        except AssertionError:
            pass
    elif tamper == "permuted_T_U":                   # test_curdleproofs.py:646-656
        from curdleproofs.util import get_permutation
        p2 = list(range(ell))
        random.shuffle(p2)
        try:
            proof.verify(crs, vec_R, vec_S, get_permutation(vec_T, p2), get_permutation(vec_U, p2), M)
            raise SystemExit("tampered proof accepted?!")
        except AssertionError:
            pass
    elif tamper in ("ipa_c_final", "same_msm_x_final", "proof_R"):
        # a malleated PROOF (not statement): the sigma-protocol asserts still pass or are not involved, and the
        # false statement is only caught by the accumulator's final MSM == A_c check (msm_accumulator.py:68)
        from py_arkworks_bls12381 import Scalar
        if tamper == "ipa_c_final":
            ipa = proof.same_perm_proof.grand_prod_proof.ipa_proof
            ipa.c_final = ipa.c_final + Scalar(1)
        elif tamper == "same_msm_x_final":
            sm = proof.same_msm_proof
            sm.x_final = sm.x_final + Scalar(1)
        else:
            proof.same_msm_proof.B_a = proof.same_msm_proof.B_a + get_random_point()
        try:
            proof.verify(crs, vec_R, vec_S, vec_T, vec_U, M)
            raise SystemExit("tampered proof accepted?!")
        except AssertionError:
            pass
    elif tamper == "wrong_M":
        try:
            proof.verify(crs, vec_R, vec_S, vec_T, vec_U, M + get_random_point())
            raise SystemExit("tampered proof accepted?!")
        except AssertionError:
            pass
    elif tamper == "wrong_k":                        # test_curdleproofs.py:662-670
        k2 = random_scalar()
        try:
            proof.verify(crs, vec_R, vec_S, [T * k2 for T in vec_T], [U * k2 for U in vec_U], M)
            raise SystemExit("tampered proof accepted?!")
        except AssertionError:
            pass
    new = RECORDS[before:]
    for r in new:
        r.update({"N": N, "seed": seed, "tamper": tamper})
    return new


def main():
    for N, seed, tamper in ((64, 1, "none"), (128, 2, "none"), (128, 3, "swap_R_S"), (64, 4, "wrong_k"),
                            (64, 5, "ipa_c_final"), (64, 6, "same_msm_x_final"), (128, 7, "ipa_c_final")):
        recs = run(N, seed, tamper)
        print(N, seed, tamper, "->", [(len(r["bases"]), r["accepts"]) for r in recs])
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "accumulator_vectors.json")
    json.dump({"generator": "tests/golden/gen_accumulator_golden.py", "records": RECORDS}, open(out, "w"))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
