#!/usr/bin/env python3
"""Proofs that carry a point OUTSIDE the prime-order subgroup, with the reference verifier's verdicts.

E(Fp): y^2 = x^3 + 4 has cofactor divisible by 3; T3 = (0, 2) has order 3.  The reference decodes points unchecked
(util.py:35-36) and asserts the same-scalar equalities (same_scalar.py:101-108) and both opening-proof equalities
(opening.py:73-76) EXACTLY, so its verdict on such a proof is deterministic; a verifier that batches those equalities
under random weights w sees  w * T3 = O  whenever 3 | w.  Cases (reference classes, unmodified, over the pure-Python
oracle backend of tests/golden/_backend.py; the provers are driven with a torsion component added where noted):

  opening  "A + T3":      A := A + T3, s recomputed for the new challenge       -> reference REJECTS (A' != A by T3)
  opening  "k_G + T3":    k_commitment := k_G + T3 and A := A + c*T3 (ground)   -> reference ACCEPTS (the defects cancel)
  shuffle  "cm_A.T_1+T3": SameScalarProof.new sends cm_A.T_1 + T3               -> reference REJECTS
  shuffle  "cancelling":  cm_T.T_1 + T3, cm_U.T_1 - T3, cm_A.T_1 - a*T3, cm_B.T_1 + a*T3 with a = alpha mod 3
                          (the same-scalar equalities hold exactly, A' = A + T_1 + U_1 has no torsion part)
                                                                                -> reference ACCEPTS, every time
plus the honest versions, and NON-CANONICAL INFINITY encodings: the wheel decodes any encoding with the infinity flag as the
identity and the reference re-serialises points before hashing, so an opening proof over the identity tracker (r_G = k_r_G =
B = identity: valid) stays valid when those identities are written as 0xC0 + junk or with the sign flag set.  (Restated from
the published ark-bls12-381 0.4 decoder in oracle/bls12_381.py; the wheel cannot run here.)
Data only -> tests/golden/torsion_vectors.json.

    python tests/golden/gen_torsion_golden.py [--backend oracle|product]
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_shuffle_golden as G  # noqa: E402  (injects the backend, imports the reference)

import curdleproofs.same_scalar as same_scalar  # noqa: E402
from curdleproofs.curdleproofs_transcript import CurdleproofsTranscript  # noqa: E402
from curdleproofs.opening import TrackerOpeningProof  # noqa: E402
from curdleproofs.util import points_projective_to_bytes  # noqa: E402
from curdleproofs.whisk_interface import IsValidWhiskOpeningProof  # noqa: E402
from py_arkworks_bls12381 import G1Point, Scalar  # noqa: E402

T3_BYTES = bytes([0x80]) + bytes(47)          # x = 0, the smaller root y = 2: a point of order 3


def t3():
    return G1Point.from_compressed_bytes_unchecked(T3_BYTES)


def pb(p):
    return bytes(G.point_projective_to_bytes(p))


def opening_case(name, torsion_on):
    k, r, blinder = G.random_scalar(), G.random_scalar(), G.random_scalar()
    r_G = G.G1 * r
    k_r_G, k_G = r_G * k, G.G1 * k
    A, B = G.G1 * blinder, r_G * blinder
    T = t3()
    if torsion_on == "A":
        A = A + T
    if torsion_on == "k_G":
        k_G = k_G + T
    tries = [None]
    if torsion_on == "k_G":                    # A := A + X with c * T3 == X: X depends on c mod 3, c depends on A: try all three
        tries = [G1Point.identity(), T, T + T]
    for X in tries:
        A_try = A if X is None else A + X
        tr = CurdleproofsTranscript(b"whisk_opening_proof")
        tr.append_list(b"tracker_opening_proof", points_projective_to_bytes([k_G, G.G1, k_r_G, r_G, A_try, B]))
        c = tr.get_and_append_challenge(b"tracker_opening_proof_challenge")
        if X is None or T * c == X:
            A = A_try
            break
    else:
        return None                              # no X fits this blinder (probability (2/3)^3): the caller retries
    s = blinder - c * k
    proof = bytes(TrackerOpeningProof(A, B, s).to_bytes())
    tracker = G.WhiskTracker(G.BLSPubkey(pb(r_G)), G.BLSPubkey(pb(k_r_G)))
    accepts = bool(IsValidWhiskOpeningProof(tracker, G.BLSPubkey(pb(k_G)), proof))
    return {"name": name, "r_G": pb(r_G).hex(), "k_r_G": pb(k_r_G).hex(), "k_commitment": pb(k_G).hex(), "proof": proof.hex(), "accepts": accepts}


ARMED = [False, 0]
_orig_gc_new = same_scalar.GroupCommitment.new.__func__
_orig_ss_new = same_scalar.SameScalarProof.new.__func__


def _gc_new(cls, crs_G, crs_H, T, r):
    cm = _orig_gc_new(cls, crs_G, crs_H, T, r)
    if ARMED[0]:
        ARMED[1] += 1
        if ARMED[1] == 1:                        # cm_A (same_scalar.py:44): send T_1 + T3
            cm.T_1 = cm.T_1 + t3()
    return cm


def _ss_new(cls, *a, **kw):
    if ARMED[0]:
        ARMED[1] = 0
        same_scalar.GroupCommitment.new = classmethod(_gc_new)
    try:
        return _orig_ss_new(cls, *a, **kw)
    finally:
        same_scalar.GroupCommitment.new = classmethod(_orig_gc_new)


same_scalar.SameScalarProof.new = classmethod(_ss_new)


# ---- cancelling torsion (the reference accepts): cm_T.T_1 += T3 and cm_U.T_1 -= T3 where the shuffle prover makes them
# (curdleproofs.py:102-107), and inside the same-scalar prover cm_A.T_1 -= a*T3, cm_B.T_1 += a*T3 with a = alpha mod 3 -- alpha
# is drawn AFTER cm_A / cm_B are absorbed, so all three guesses of a are tried on copies of the transcript.
CANCEL = {"on": False, "calls": 0, "guess": 0}
import curdleproofs.curdleproofs as cp_mod  # noqa: E402

_orig_cp_new = cp_mod.CurdleProofsProof.new.__func__


class NoGuessFits(Exception):
    pass


def _gc_new_cancel(cls, crs_G, crs_H, T, r):
    cm = _orig_gc_new(cls, crs_G, crs_H, T, r)
    if not CANCEL["on"]:                          # the verifier's own GroupCommitment.new calls (same_scalar.py:101-106)
        return cm
    CANCEL["calls"] += 1
    n, T3 = CANCEL["calls"], t3()
    if n == 1:
        cm.T_1 = cm.T_1 + T3                      # cm_T
    elif n == 2:
        cm.T_1 = cm.T_1 - T3                      # cm_U
    elif n % 2 == 1:                              # cm_A of a same-scalar attempt
        for _ in range(CANCEL["guess"]):
            cm.T_1 = cm.T_1 - T3
    else:                                         # cm_B
        for _ in range(CANCEL["guess"]):
            cm.T_1 = cm.T_1 + T3
    return cm


def _ss_new_cancel(cls, crs_G_t, crs_G_u, crs_H, R, S, cm_T, cm_U, k, r_t, r_u, transcript):
    import copy

    t0 = copy.deepcopy(transcript)
    for guess in range(3):
        CANCEL["guess"] = guess
        tr = copy.deepcopy(t0)
        proof = _orig_ss_new(cls, crs_G_t=crs_G_t, crs_G_u=crs_G_u, crs_H=crs_H, R=R, S=S, cm_T=cm_T, cm_U=cm_U, k=k, r_t=r_t, r_u=r_u, transcript=tr)
        CANCEL["on"] = False
        try:
            proof.verify(crs_G_t, crs_G_u, crs_H, R, S, cm_T, cm_U, copy.deepcopy(t0))      # same_scalar.py:71-108, exact equalities
        except AssertionError:
            continue
        finally:
            CANCEL["on"] = True
        transcript.__dict__.update(tr.__dict__)   # the caller's transcript continues from the accepted attempt
        return proof
    raise NoGuessFits()


def cancelling_shuffle_case(name, ell, crs, want_nonzero=False):
    pre = G.make_trackers(ell)
    while True:
        CANCEL.update(on=True, calls=0, guess=0)
        same_scalar.GroupCommitment.new = classmethod(_gc_new_cancel)
        same_scalar.SameScalarProof.new = classmethod(_ss_new_cancel)
        try:
            post, proof = G.GenerateWhiskShuffleProof(crs, pre)
            if want_nonzero and CANCEL["guess"] == 0:
                continue                              # this case shall carry torsion on cm_A / cm_B too
            break
        except NoGuessFits:
            continue
        finally:
            same_scalar.GroupCommitment.new = classmethod(_orig_gc_new)
            same_scalar.SameScalarProof.new = classmethod(_ss_new)
            CANCEL["on"] = False
    pre_r, pre_k = G.cat(pre)
    post_r, post_k = G.cat(post)
    accepts = bool(G.IsValidWhiskShuffleProof(crs, pre, post, proof))
    return {"name": name, "pre_r": pre_r.hex(), "pre_k": pre_k.hex(), "post_r": post_r.hex(), "post_k": post_k.hex(),
            "proof": bytes(proof).hex(), "accepts": accepts, "alpha_mod_3": CANCEL["guess"]}


def shuffle_case(name, ell, crs, torsion):
    pre = G.make_trackers(ell)
    ARMED[0] = torsion
    try:
        post, proof = G.GenerateWhiskShuffleProof(crs, pre)
    finally:
        ARMED[0] = False
    pre_r, pre_k = G.cat(pre)
    post_r, post_k = G.cat(post)
    accepts = bool(G.IsValidWhiskShuffleProof(crs, pre, post, proof))
    return {"name": name, "pre_r": pre_r.hex(), "pre_k": pre_k.hex(), "post_r": post_r.hex(), "post_k": post_k.hex(),
            "proof": bytes(proof).hex(), "accepts": accepts}


def identity_tracker_case(name, enc_rG, enc_krG, enc_B, enc_A=None):
    """Opening proof over r_G = identity: k_r_G = identity, B = identity; the three identities encoded as given."""
    k, blinder = G.random_scalar(), G.random_scalar()
    Z = G1Point.identity()
    k_G, A = G.G1 * k, G.G1 * blinder
    tr = CurdleproofsTranscript(b"whisk_opening_proof")
    tr.append_list(b"tracker_opening_proof", points_projective_to_bytes([k_G, G.G1, Z, Z, A, Z]))
    c = tr.get_and_append_challenge(b"tracker_opening_proof_challenge")
    s = blinder - c * k
    proof = (enc_A or pb(A)) + enc_B + bytes(s.to_le_bytes())
    tracker = G.WhiskTracker(G.BLSPubkey(enc_rG), G.BLSPubkey(enc_krG))
    accepts = bool(IsValidWhiskOpeningProof(tracker, G.BLSPubkey(pb(k_G)), proof))
    return {"name": name, "r_G": enc_rG.hex(), "k_r_G": enc_krG.hex(), "k_commitment": pb(k_G).hex(), "proof": proof.hex(), "accepts": accepts}


def main():
    random.seed(333)
    opening = []
    for name, where in (("honest", None), ("A + T3 (s recomputed)", "A"), ("k_G + T3, A + c*T3", "k_G"), ("honest 2", None),
                        ("k_G + T3, A + c*T3 (2)", "k_G"), ("A + T3 (2)", "A")):
        case = None
        while case is None:
            case = opening_case(name, where)
        opening.append(case)
    canon = bytes([0xC0]) + bytes(47)
    junk = bytes([0xC0]) + bytes(range(1, 48))
    signed = bytes([0xE0]) + bytes(47)
    tail = bytes([0xC0]) + bytes(46) + b"\x01"
    opening += [identity_tracker_case("identity tracker, canonical encodings", canon, canon, canon),
                identity_tracker_case("identity tracker, non-canonical infinity encodings", junk, signed, tail),
                identity_tracker_case("identity tracker, A := non-canonical infinity", canon, junk, canon, enc_A=signed)]
    ell = 12
    crs = G.CurdleproofsCrs.new(ell, G.N_BLINDERS)
    shuffle = {"ell": ell, "crs": bytes(crs.to_bytes()).hex(),
               "cases": [shuffle_case("honest", ell, crs, False), shuffle_case("cm_A.T_1 + T3", ell, crs, True),
                         shuffle_case("honest 2", ell, crs, False), shuffle_case("cm_A.T_1 + T3 (2)", ell, crs, True),
                         cancelling_shuffle_case("cancelling T3 on cm_T / cm_U / cm_A / cm_B", ell, crs),
                         cancelling_shuffle_case("cancelling T3 on cm_T / cm_U / cm_A / cm_B (2)", ell, crs, want_nonzero=True)]}
    out = {"generator": "tests/golden/gen_torsion_golden.py (reference classes; G1Point/Scalar = %s)" % G.BACKEND_MODULE,
           "backend": G.BACKEND_MODULE, "t3": T3_BYTES.hex(), "opening": opening, "shuffle": shuffle,
           "unpinned": "the 'identity tracker' cases rest on the decoding rule that ANY encoding with the infinity flag is the identity: restated from the "
                       "published decoder of the wheel's crate (ark-bls12-381 0.4 read_g1_compressed), not pinned by a vector of the real "
                       "py_arkworks_bls12381 0.3.5 wheel (it cannot run in the build container)"}
    path = G._backend.out_path("torsion_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("opening:", [(c["name"], c["accepts"]) for c in opening])
    print("shuffle:", [(c["name"], c["accepts"]) for c in shuffle["cases"]])
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
