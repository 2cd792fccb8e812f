#!/usr/bin/env python3
"""Protocol-shaped inputs and outputs of the reference PROVER's hot loops, for the GPU fold / map kernels.

Runs the reference's GenerateWhiskShuffleProof (whisk_interface.py:111-144), unmodified, over the CPU-oracle backend
(tests/golden/_backend.py) at ell = 28 (+4 blinders = 32) and records, through wrappers around the reference's own functions:

  * shuffle_permute_and_commit_input (curdleproofs.py:301-321): vec_R, vec_S, permutation, k -> vec_T, vec_U, M, blinders
  * IPA.new (ipa.py:75-153): crs_G_vec, crs_G_prime_vec, H = crs_H * beta, the blinded vec_c / vec_d as they enter the halving
    loop (:117), the ipa_gamma challenges, and the proof's vec_L_C / vec_R_C / vec_L_D / vec_R_D / c_final / d_final
  * SameMSMProof.new (same_msm.py:50-143): crs_G_vec, vec_T, vec_U, the blinded vec_x (:90-91), the same_msm_gamma challenges,
    the proof's six L/R vectors and x_final
  * the grand-product base change (grand_prod.py:64-71): crs vec_G / vec_H, beta^-1, and G' | H' (= IPA.new's crs_G_prime_vec)
Data only -> tests/golden/prover_vectors.json.

    python tests/golden/gen_prover_golden.py [--backend oracle|product]
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_shuffle_golden as G  # noqa: E402  (injects the backend, imports the reference, records challenges)

import curdleproofs.ipa as ipa_mod  # noqa: E402
import curdleproofs.same_msm as same_msm_mod  # noqa: E402
import curdleproofs.whisk_interface as wi  # noqa: E402

REC = {}
pt = lambda p: bytes(G.point_projective_to_bytes(p)).hex()
fr = lambda s: bytes(s.to_le_bytes()).hex()


def challenges_since(mark, label):
    return [v for lab, v in G.CHALLENGES[mark:] if lab == label]


_ipa_new = ipa_mod.IPA.new.__func__


def ipa_new(cls, crs_G_vec, crs_G_prime_vec, crs_H, C, D, z, vec_c, vec_d, transcript):
    mark = len(G.CHALLENGES)
    g, gp = list(crs_G_vec), list(crs_G_prime_vec)
    proof = _ipa_new(cls, crs_G_vec, crs_G_prime_vec, crs_H, C, D, z, vec_c, vec_d, transcript)
    beta = G.Scalar.from_le_bytes(bytes.fromhex(challenges_since(mark, "ipa_beta")[0]))
    REC["ipa"] = {"crs_G_vec": [pt(p) for p in g], "crs_G_prime_vec": [pt(p) for p in gp], "H": pt(crs_H * beta),
                  "vec_c": [fr(s) for s in vec_c], "vec_d": [fr(s) for s in vec_d],          # mutated in place: the blinded vectors (ipa.py:107-109)
                  "gammas": challenges_since(mark, "ipa_gamma"),
                  "vec_L_C": [pt(p) for p in proof.vec_L_C], "vec_R_C": [pt(p) for p in proof.vec_R_C],
                  "vec_L_D": [pt(p) for p in proof.vec_L_D], "vec_R_D": [pt(p) for p in proof.vec_R_D],
                  "c_final": fr(proof.c_final), "d_final": fr(proof.d_final)}
    return proof


ipa_mod.IPA.new = classmethod(ipa_new)
_sm_new = same_msm_mod.SameMSMProof.new.__func__


def sm_new(cls, crs_G_vec, A, Z_t, Z_u, vec_T, vec_U, vec_x, transcript):
    mark = len(G.CHALLENGES)
    g, t, u = list(crs_G_vec), list(vec_T), list(vec_U)
    proof = _sm_new(cls, crs_G_vec, A, Z_t, Z_u, vec_T, vec_U, vec_x, transcript)
    REC["same_msm"] = {"crs_G_vec": [pt(p) for p in g], "vec_T": [pt(p) for p in t], "vec_U": [pt(p) for p in u],
                       "vec_x": [fr(s) for s in vec_x],                                      # mutated in place: blinded (same_msm.py:90-91)
                       "gammas": challenges_since(mark, "same_msm_gamma"),
                       **{k: [pt(p) for p in getattr(proof, k)] for k in ("vec_L_A", "vec_L_T", "vec_L_U", "vec_R_A", "vec_R_T", "vec_R_U")},
                       "x_final": fr(proof.x_final)}
    return proof


same_msm_mod.SameMSMProof.new = classmethod(sm_new)
_spci = wi.shuffle_permute_and_commit_input


def spci(crs, vec_R, vec_S, permutation, k):
    state = random.getstate()
    out = _spci(crs, vec_R, vec_S, permutation, k)
    vec_T, vec_U, M, blinders = out
    REC["permute_commit"] = {"vec_R": [pt(p) for p in vec_R], "vec_S": [pt(p) for p in vec_S], "permutation": list(permutation), "k": fr(k),
                             "vec_T": [pt(p) for p in vec_T], "vec_U": [pt(p) for p in vec_U], "M": pt(M), "blinders": [fr(b) for b in blinders]}
    REC["_rng_state_before_blinders"] = state
    return out


wi.shuffle_permute_and_commit_input = spci


def main():
    ell = 28
    random.seed(4711)
    crs = G.CurdleproofsCrs.new(ell, G.N_BLINDERS)
    pre = G.make_trackers(ell)
    del G.CHALLENGES[:]
    post, proof = G.GenerateWhiskShuffleProof(crs, pre)
    assert G.IsValidWhiskShuffleProof(crs, pre, post, proof)
    REC.pop("_rng_state_before_blinders")
    beta = G.Scalar.from_le_bytes(bytes.fromhex([v for lab, v in G.CHALLENGES if lab == "gprod_beta"][0]))
    REC["grand_product_bases"] = {"vec_G": [pt(p) for p in crs.vec_G], "vec_H": [pt(p) for p in crs.vec_H], "beta_inv": fr(beta.inverse()),
                                  "G_prime_H_prime": REC["ipa"]["crs_G_prime_vec"]}
    out = {"generator": "tests/golden/gen_prover_golden.py (reference prover; G1Point/Scalar = %s)" % G.BACKEND_MODULE, "backend": G.BACKEND_MODULE,
           "ell": ell, "crs": bytes(crs.to_bytes()).hex(), **REC}
    path = G._backend.out_path("prover_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print({k: (list(v.keys()) if isinstance(v, dict) else v) for k, v in out.items() if k not in ("crs",)})
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
