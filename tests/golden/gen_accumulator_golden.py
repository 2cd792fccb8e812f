#!/usr/bin/env python3
"""Record REAL protocol data for the hot path: run the reference's own shuffle prover and verifier
(/root/reference/curdleproofs, imported unmodified in the build container with our G1Point/Scalar module
standing in for the missing Rust wheel) on seeded inputs, and dump what its `MSMAccumulator.verify()`
(msm_accumulator.py:60-68) feeds to `compute_MSM` -- the 5*ell+7 unique bases, the merged scalars -- together
with the left-hand side A_c it is compared with, for valid proofs and for the reference's own tampered cases
(test_curdleproofs.py:643-670).  Only data is written: tests/golden/accumulator_vectors.json
(compressed points 48 B hex, scalars 32 B LE hex).  The GPU tests recompute every MSM and compare with A_c.

The curve arithmetic during generation is the pure-Python CPU oracle (tests/golden/_backend.py; the wheel cannot run
here), so nothing in the file comes out of product arithmetic; `--backend product` regenerates it over the product's host
C++ and must give the same bytes (tests/test_golden_backends.py).  Not an independent pin of the group law (that is
tests/test_oracle_kat.py).

Also recorded ("sequences"): the complete `accumulate_check` call sequence of one verify (msm_accumulator.py:37-58) --
per call the left-hand side C, every (base, scalar) pair as passed (identity bases included) and the random factor the
reference drew -- followed by the accumulator's final state.  tests replay it through the product's MSMAccumulator on the
GPU and through oracle.MSMAccumulator on the CPU.

    python tests/golden/gen_accumulator_golden.py
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _backend  # noqa: E402

BACKEND_MODULE = _backend.inject()

import curdleproofs.msm_accumulator as ref_acc  # noqa: E402
from curdleproofs.crs import CurdleproofsCrs  # noqa: E402
from curdleproofs.curdleproofs import N_BLINDERS, CurdleProofsProof, shuffle_permute_and_commit_input  # noqa: E402
from curdleproofs.util import get_random_point, random_scalar  # noqa: E402

RECORDS = []
CALLS = None                      # list of accumulate_check calls while a sequence is being recorded
_orig_verify = ref_acc.MSMAccumulator.verify
_orig_accumulate = ref_acc.MSMAccumulator.accumulate_check
_orig_random_scalar = ref_acc.random_scalar
_last_rho = []


def _recording_random_scalar():
    r = _orig_random_scalar()
    _last_rho.append(r)
    return r


ref_acc.random_scalar = _recording_random_scalar


def recording_accumulate(self, C, bases, scalars):
    bases, scalars = list(bases), list(scalars)
    del _last_rho[:]
    _orig_accumulate(self, C, bases, scalars)
    if CALLS is not None:
        assert len(_last_rho) == 1            # exactly one draw per call (msm_accumulator.py:43)
        CALLS.append({"C": bytes(C.to_compressed_bytes()).hex(),
                      "bases": [bytes(b.to_compressed_bytes()).hex() for b in bases],
                      "scalars": [bytes(x.to_le_bytes()).hex() for x in scalars],
                      "rho": bytes(_last_rho[0].to_le_bytes()).hex()})


ref_acc.MSMAccumulator.accumulate_check = recording_accumulate


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


def record_sequence(N, seed, tamper):
    global CALLS
    CALLS = []
    recs = run(N, seed, tamper)
    calls, CALLS = CALLS, None
    assert len(recs) == 1
    final = recs[0]
    return {"N": N, "seed": seed, "tamper": tamper, "calls": calls,
            "final": {"bases": final["bases"], "scalars": final["scalars"], "A_c": final["A_c"], "accepts": final["accepts"]}}


def main():
    sequences = [record_sequence(128, 21, "none"), record_sequence(64, 22, "same_msm_x_final")]
    for q in sequences:
        print("sequence", q["N"], q["tamper"], [len(c["bases"]) for c in q["calls"]], "->", len(q["final"]["bases"]), q["final"]["accepts"])
    del RECORDS[:]
    for N, seed, tamper in ((64, 1, "none"), (128, 2, "none"), (128, 3, "swap_R_S"), (64, 4, "wrong_k"),
                            (64, 5, "ipa_c_final"), (64, 6, "same_msm_x_final"), (128, 7, "ipa_c_final")):
        recs = run(N, seed, tamper)
        print(N, seed, tamper, "->", [(len(r["bases"]), r["accepts"]) for r in recs])
    out = _backend.out_path("accumulator_vectors.json")
    json.dump({"generator": "tests/golden/gen_accumulator_golden.py (G1Point/Scalar = %s)" % BACKEND_MODULE, "backend": BACKEND_MODULE,
               "records": RECORDS, "sequences": sequences}, open(out, "w"), separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
