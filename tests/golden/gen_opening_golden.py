#!/usr/bin/env python3
"""Record Whisk tracker-opening proofs and the reference verifier's verdicts (+ its Fiat-Shamir challenge).

Runs the reference's own entry points (/root/reference/curdleproofs/curdleproofs/whisk_interface.py:
GenerateWhiskTrackerProof :172-190, IsValidWhiskOpeningProof :147-169), imported unmodified in the build container
with a stand-in for the missing Rust wheel (tests/golden/_backend.py: the pure-Python CPU oracle by default), on seeded inputs.  Data only ->
tests/golden/opening_vectors.json: per case the tracker, k_commitment, proof bytes, the challenge the reference
verifier drew, and tampered variants each with the verdict IsValidWhiskOpeningProof returned.

    python tests/golden/gen_opening_golden.py
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_shuffle_golden as G  # noqa: E402  (injects the backend, imports the reference, records challenges)

from curdleproofs.whisk_interface import GenerateWhiskTrackerProof, IsValidWhiskOpeningProof  # noqa: E402


def main():
    random.seed(4242)
    cases = []
    other = bytes(G.point_projective_to_bytes(G.G1 * G.Scalar(777)))
    identity = b"\xc0" + bytes(47)
    for _ in range(6):
        k, r = G.random_scalar(), G.random_scalar()
        r_G = G.G1 * r
        tracker = G.WhiskTracker(G.BLSPubkey(G.point_projective_to_bytes(r_G)), G.BLSPubkey(G.point_projective_to_bytes(r_G * k)))
        k_commitment = G.BLSPubkey(G.point_projective_to_bytes(G.G1 * k))
        proof = bytes(GenerateWhiskTrackerProof(tracker, k))
        assert len(proof) == 128
        del G.CHALLENGES[:]
        assert IsValidWhiskOpeningProof(tracker, k_commitment, proof)
        challenge = G.CHALLENGES[-1][1]
        base = {"r_G": bytes(tracker.r_G), "k_r_G": bytes(tracker.k_r_G), "k_commitment": bytes(k_commitment), "proof": proof}
        variants = []

        def add(name, **edits):
            b = dict(base)
            b.update(edits)
            ok = bool(IsValidWhiskOpeningProof(G.WhiskTracker(G.BLSPubkey(b["r_G"]), G.BLSPubkey(b["k_r_G"])), G.BLSPubkey(b["k_commitment"]), b["proof"]))
            variants.append({"name": name, "edits": {k2: v.hex() for k2, v in edits.items()}, "accepts": ok})

        s_plus = ((int.from_bytes(proof[96:], "little") + 1) % G.FR_MODULUS).to_bytes(32, "little")
        add("no edit")
        add("trailing bytes", proof=proof + b"\x07\x07")
        add("A := other", proof=other + proof[48:])
        add("B := other", proof=proof[:48] + other + proof[96:])
        add("s += 1", proof=proof[:96] + s_plus)
        add("s := r (non-canonical)", proof=proof[:96] + G.FR_MODULUS.to_bytes(32, "little"))
        add("A := identity", proof=identity + proof[48:])
        add("B := bad flags", proof=proof[:48] + b"\x00" + proof[49:])
        add("k_commitment := other", k_commitment=other)
        add("r_G := other", r_G=other)
        add("k_r_G := other", k_r_G=other)
        add("swap r_G <-> k_r_G", r_G=base["k_r_G"], k_r_G=base["r_G"])
        add("truncated proof", proof=proof[:-1])
        cases.append({**{k2: v.hex() for k2, v in base.items()}, "challenge": challenge, "variants": variants})
    path = G._backend.out_path("opening_vectors.json")
    with open(path, "w") as f:
        json.dump({"generator": "tests/golden/gen_opening_golden.py (reference whisk_interface; G1Point/Scalar = %s)" % G.BACKEND_MODULE,
                   "backend": G.BACKEND_MODULE, "cases": cases}, f, separators=(",", ":"))
    print([(v["name"], v["accepts"]) for v in cases[0]["variants"]])
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
