#!/usr/bin/env python3
"""Record wire-format shuffle proofs and the reference verifier's verdicts + Fiat-Shamir challenges.

Runs the reference's own Whisk entry points (/root/reference/curdleproofs/curdleproofs/whisk_interface.py:
GenerateWhiskShuffleProof :113-144, IsValidWhiskShuffleProof :72-109), imported unmodified in the build container
with a stand-in for the missing Rust wheel (tests/golden/_backend.py: the pure-Python CPU oracle by default, so no
byte of the fixture comes out of product arithmetic; `--backend product` for the byte-identity cross-check) and the
reference's own pure-Python Merlin, on seeded inputs.  Written to tests/golden/shuffle_vectors.json -- data only:

  per case: ell, CRS bytes (crs.py:92-101), pre/post tracker encodings, proof bytes, every challenge the reference
  verifier drew (label + 32 LE bytes, in order), and a list of tampered variants (byte edits of the proof or the
  trackers) each with the verdict IsValidWhiskShuffleProof returned.

The batch verifier (curdleproofs_pie_amd/shuffle_verifier.py over csrc/shuffle_verify.cpp) must reproduce the
challenges bit-for-bit and the verdicts exactly; tests/test_shuffle_verifier.py (CPU) and
tests/test_shuffle_verifier_gpu.py check that without the reference present.

    python tests/golden/gen_shuffle_golden.py
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _backend  # noqa: E402

BACKEND_MODULE = _backend.inject()

from curdleproofs.crs import CurdleproofsCrs  # noqa: E402
from curdleproofs.curdleproofs_transcript import CurdleproofsTranscript  # noqa: E402
from curdleproofs.util import BLSPubkey, G1, point_projective_to_bytes, random_scalar  # noqa: E402
from curdleproofs.whisk_interface import GenerateWhiskShuffleProof, IsValidWhiskShuffleProof, WhiskTracker  # noqa: E402
from py_arkworks_bls12381 import Scalar  # noqa: E402

N_BLINDERS = 4
FR_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001      # util.py:7
CHALLENGES = []
_orig = CurdleproofsTranscript.get_and_append_challenge


def _recording(self, label):
    f = _orig(self, label)
    CHALLENGES.append([label.decode(), bytes(f.to_le_bytes()).hex()])
    return f


CurdleproofsTranscript.get_and_append_challenge = _recording


def make_trackers(n):
    out = []
    for _ in range(n):
        k, r = random_scalar(), random_scalar()
        r_G = G1 * r
        out.append(WhiskTracker(BLSPubkey(point_projective_to_bytes(r_G)), BLSPubkey(point_projective_to_bytes(r_G * k))))
    return out


def cat(trackers):
    return b"".join(bytes(t.r_G) for t in trackers), b"".join(bytes(t.k_r_G) for t in trackers)


def split(r_bytes, k_bytes):
    n = len(r_bytes) // 48
    return [WhiskTracker(BLSPubkey(r_bytes[48 * i: 48 * i + 48]), BLSPubkey(k_bytes[48 * i: 48 * i + 48])) for i in range(n)]


def apply_edits(bufs, edits):
    bufs = {k: bytearray(v) for k, v in bufs.items()}
    for which, off, hexbytes in edits:
        b = bytes.fromhex(hexbytes)
        bufs[which][off: off + len(b)] = b
    return {k: bytes(v) for k, v in bufs.items()}


def proof_offsets(lg):
    """Byte offsets of the named fields of WhiskShuffleProof.to_bytes (whisk_interface.py:56-61 and the nested to_bytes)."""
    names = ["M", "A", "T_1", "T_2", "U_1", "U_2", "R", "S", "B", "C", ("r_p",), "B_c", "B_d"]
    names += [f"L_C{j}" for j in range(lg)] + [f"R_C{j}" for j in range(lg)] + [f"L_D{j}" for j in range(lg)] + [f"R_D{j}" for j in range(lg)]
    names += [("c_final",), ("d_final",), "cmA_1", "cmA_2", "cmB_1", "cmB_2", ("z_k",), ("z_t",), ("z_u",), "B_a", "B_t", "B_u"]
    for v in ("L_A", "L_T", "L_U", "R_A", "R_T", "R_U"):
        names += [f"{v}{j}" for j in range(lg)]
    names += [("x_final",)]
    off, out = 0, {}
    for nm in names:
        if isinstance(nm, tuple):
            out[nm[0]] = (off, 32)
            off += 32
        else:
            out[nm] = (off, 48)
            off += 48
    return out, off


def run_case(ell, seed, n_variants):
    random.seed(seed)
    crs = CurdleproofsCrs.new(ell, N_BLINDERS)
    pre = make_trackers(ell)
    post, proof = GenerateWhiskShuffleProof(crs, pre)
    proof = bytes(proof)
    lg = (ell + N_BLINDERS).bit_length() - 1
    offs, total = proof_offsets(lg)
    assert total == len(proof), (total, len(proof))
    pre_r, pre_k = cat(pre)
    post_r, post_k = cat(post)
    bufs = {"proof": proof, "pre_r": pre_r, "pre_k": pre_k, "post_r": post_r, "post_k": post_k}

    def verdict(b):
        return bool(IsValidWhiskShuffleProof(crs, split(b["pre_r"], b["pre_k"]), split(b["post_r"], b["post_k"]), b["proof"]))

    del CHALLENGES[:]
    assert verdict(bufs)
    challenges = list(CHALLENGES)

    other_point = bytes(point_projective_to_bytes(G1 * Scalar(0xC0FFEE + seed))).hex()
    identity = (b"\xc0" + b"\x00" * 47).hex()
    variants = []

    def add(name, edits):
        variants.append({"name": name, "edits": edits, "accepts": verdict(apply_edits(bufs, edits))})

    rng = random.Random(seed * 7 + 1)
    fields = list(offs.keys())
    rng.shuffle(fields)
    for nm in fields[:n_variants]:                                  # every kind of field gets hit across the cases
        off, size = offs[nm]
        if size == 48:
            add(f"proof.{nm} := other point", [["proof", off, other_point]])
        else:
            v = (int.from_bytes(proof[off: off + 32], "little") + 1) % FR_MODULUS
            add(f"proof.{nm} += 1", [["proof", off, v.to_bytes(32, "little").hex()]])
    add("proof.x_final := r (non-canonical)", [["proof", offs["x_final"][0], FR_MODULUS.to_bytes(32, "little").hex()]])
    add("proof.L_A0 := identity", [["proof", offs["L_A0"][0], identity]])
    add("proof.C := bad flags", [["proof", offs["C"][0], "00" + proof[offs["C"][0] + 1: offs["C"][0] + 48].hex()]])
    add("proof.B_c := x not on curve", [["proof", offs["B_c"][0], (0x80).to_bytes(1, "big").hex() + (5).to_bytes(47, "big").hex()]])
    add("post_r[0] := identity", [["post_r", 0, identity]])
    add("post_r[1] := other point", [["post_r", 48, other_point]])
    add("pre_k[0] := other point", [["pre_k", 0, other_point]])
    add("swap pre_r <-> pre_k", [["pre_r", 0, pre_k.hex()], ["pre_k", 0, pre_r.hex()]])
    add("post_k[last] <-> post_k[0]", [["post_k", 0, post_k[-48:].hex()], ["post_k", len(post_k) - 48, post_k[:48].hex()]])
    add("no edit", [])
    return {
        "ell": ell, "seed": seed, "crs": bytes(crs.to_bytes()).hex(),
        "pre_r": pre_r.hex(), "pre_k": pre_k.hex(), "post_r": post_r.hex(), "post_k": post_k.hex(),
        "proof": proof.hex(), "challenges": challenges, "variants": variants,
    }


def main():
    cases = [run_case(4, 11, 12), run_case(12, 12, 16), run_case(28, 13, 10), run_case(60, 14, 8), run_case(124, 15, 8), run_case(124, 16, 0)]
    out = {"generator": "tests/golden/gen_shuffle_golden.py (reference whisk_interface; G1Point/Scalar = %s)" % BACKEND_MODULE,
           "backend": BACKEND_MODULE, "cases": cases}
    path = _backend.out_path("shuffle_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    for c in cases:
        print("ell", c["ell"], "variants", [(v["name"], v["accepts"]) for v in c["variants"]])
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
