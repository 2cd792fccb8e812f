#!/usr/bin/env python3
"""BASELINE configs 3 / 5 fixture: DISTINCT Whisk shuffle proofs (ell = 124 + 4 blinders = 128) over one CRS.

Runs the reference's own prover and verifier (/root/reference/curdleproofs/curdleproofs/whisk_interface.py:
GenerateWhiskShuffleProof :111-144, IsValidWhiskShuffleProof :72-87), imported unmodified in the build container with the
stand-in of tests/golden/_backend.py for the missing Rust wheel (the pure-Python CPU oracle by default), one seeded worker
process per proof.  Every proof has its own trackers, permutation and k; every one is accepted by the reference verifier
before it is written.  A handful of tampered variants (byte edits of a proof or its trackers) carry the verdict
IsValidWhiskShuffleProof returned for them.  Data only:

  shuffle_batch_ell124.bin    crs (133 x 48 B) | count x ( pre_r | pre_k | post_r | post_k | proof )
  shuffle_batch_ell124.json   sizes, sha256 of the .bin, seeds, the tampered variants {base, edits, accepts}

bench.py tiles these to a batch of 1024 (fresh random weights per slot); tests/test_shuffle_batch_gpu.py pushes such a
batch with tampered proofs at known slots through the GPU verifier.

    python tests/golden/gen_shuffle_batch.py [--backend oracle|product] [--count 64] [--workers 8] [--out DIR]

EXTENSION to 1024 distinct proofs ("1 024 / 16 384 distinct proofs"):
    python tests/golden/gen_shuffle_batch.py --first 64 --count 960 --workers 5
writes shuffle_batch_ell124_more.{bin,json}: proofs 64 .. 1023 of the SAME seeded sequence (seed 9100 + i, same CRS), records
only, made like the first 64 by the reference prover over the oracle backend (35 minutes on five cores; `--backend product`
writes the very same 27 MB in 5 minutes -- tests/test_golden_backends.py re-makes records of it over both backends).  Every proof
is accepted by the reference verifier before it is written.
"""
import hashlib
import json
import os
import random
import sys
from concurrent.futures import ProcessPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _backend  # noqa: E402

ELL, N_BLINDERS = 124, 4
CRS_SEED, PROOF_SEED = 9000, 9100


def _setup():
    _backend.inject()
    import gen_shuffle_golden as G
    return G


def make_crs():
    G = _setup()
    random.seed(CRS_SEED)
    return bytes(G.CurdleproofsCrs.new(ELL, N_BLINDERS).to_bytes())


def make_proof(args):
    i, crs_bytes = args
    G = _setup()
    from curdleproofs.util import BufReader

    crs = G.CurdleproofsCrs.from_bytes(BufReader(crs_bytes), ELL, N_BLINDERS)
    random.seed(PROOF_SEED + i)
    pre = G.make_trackers(ELL)
    post, proof = G.GenerateWhiskShuffleProof(crs, pre)
    assert G.IsValidWhiskShuffleProof(crs, pre, post, proof)
    pre_r, pre_k = G.cat(pre)
    post_r, post_k = G.cat(post)
    return i, pre_r, pre_k, post_r, post_k, bytes(proof)


def judge(args):
    """Reference verdict of one tampered variant."""
    crs_bytes, bufs, edits = args
    G = _setup()
    from curdleproofs.util import BufReader

    crs = G.CurdleproofsCrs.from_bytes(BufReader(crs_bytes), ELL, N_BLINDERS)
    b = G.apply_edits(bufs, edits)
    return bool(G.IsValidWhiskShuffleProof(crs, G.split(b["pre_r"], b["pre_k"]), G.split(b["post_r"], b["post_k"]), b["proof"]))


def extend(first, count, workers, out_dir, crs_bytes):
    with ProcessPoolExecutor(workers) as ex:
        proofs = sorted(ex.map(make_proof, [(i, crs_bytes) for i in range(first, first + count)], chunksize=4))
    blob = b"".join(b"".join(p[1:]) for p in proofs)
    with open(os.path.join(out_dir, "shuffle_batch_ell124_more.bin"), "wb") as f:
        f.write(blob)
    meta = {"generator": "tests/golden/gen_shuffle_batch.py --first %d (reference whisk_interface; G1Point/Scalar = %s)" % (first, _backend.inject()),
            "backend": _backend.inject(), "ell": ELL, "n_blinders": N_BLINDERS, "first": first, "count": count,
            "record_bytes": len(blob) // count, "crs_seed": CRS_SEED, "proof_seed_base": PROOF_SEED, "sha256": hashlib.sha256(blob).hexdigest(),
            "note": "proofs first .. first+count-1 of the seeded sequence of shuffle_batch_ell124.bin, same CRS; every one accepted by the "
                    "reference's IsValidWhiskShuffleProof when it was made"}
    with open(os.path.join(out_dir, "shuffle_batch_ell124_more.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", len(blob), "bytes:", count, "more distinct proofs")


def main():
    count = int(_backend._arg("--count", "64"))
    workers = int(_backend._arg("--workers", "8"))
    out_dir = _backend.OUT or os.path.dirname(os.path.abspath(__file__))
    crs_bytes = make_crs()
    first = int(_backend._arg("--first", "0"))
    if first:
        return extend(first, count, workers, out_dir, crs_bytes)
    with ProcessPoolExecutor(workers) as ex:
        proofs = sorted(ex.map(make_proof, [(i, crs_bytes) for i in range(count)]))
        G = _setup()
        lg = (ELL + N_BLINDERS).bit_length() - 1
        offs, total = G.proof_offsets(lg)
        assert all(len(p[5]) == total for p in proofs)
        other = bytes(G.point_projective_to_bytes(G.G1 * G.Scalar(0xBEEF))).hex()
        identity = (b"\xc0" + bytes(47)).hex()
        fr_plus = lambda pr, nm: ((int.from_bytes(pr[offs[nm][0]: offs[nm][0] + 32], "little") + 1) % G.FR_MODULUS).to_bytes(32, "little").hex()
        plan = []                                              # (base proof, name, edits)
        rng = random.Random(77)
        bases = rng.sample(range(count), 12)
        pr = lambda b: proofs[b][5]
        plan.append((bases[0], "proof.x_final += 1", [["proof", offs["x_final"][0], fr_plus(pr(bases[0]), "x_final")]]))
        plan.append((bases[1], "proof.c_final += 1", [["proof", offs["c_final"][0], fr_plus(pr(bases[1]), "c_final")]]))
        plan.append((bases[2], "proof.B_a := other point", [["proof", offs["B_a"][0], other]]))
        plan.append((bases[3], "proof.z_k += 1 (same-scalar argument)", [["proof", offs["z_k"][0], fr_plus(pr(bases[3]), "z_k")]]))
        plan.append((bases[4], "proof.T_1 := other point (same-scalar argument)", [["proof", offs["T_1"][0], other]]))
        plan.append((bases[5], "post_r[3] := other point", [["post_r", 3 * 48, other]]))
        plan.append((bases[6], "swap pre_r <-> pre_k", [["pre_r", 0, proofs[bases[6]][2].hex()], ["pre_k", 0, proofs[bases[6]][1].hex()]]))
        plan.append((bases[7], "post_r[0] := identity", [["post_r", 0, identity]]))
        plan.append((bases[8], "proof.M := bad flags", [["proof", 0, "00"]]))
        plan.append((bases[9], "proof.x_final := r (non-canonical)", [["proof", offs["x_final"][0], G.FR_MODULUS.to_bytes(32, "little").hex()]]))
        plan.append((bases[10], "proof of another instance", [["proof", 0, pr(bases[11]).hex()]]))
        plan.append((bases[11], "no edit", []))
        keys = ("pre_r", "pre_k", "post_r", "post_k", "proof")
        verdicts = list(ex.map(judge, [(crs_bytes, dict(zip(keys, proofs[b][1:])), edits) for b, _, edits in plan]))
    blob = crs_bytes + b"".join(b"".join(p[1:]) for p in proofs)
    with open(os.path.join(out_dir, "shuffle_batch_ell124.bin"), "wb") as f:
        f.write(blob)
    meta = {
        "generator": "tests/golden/gen_shuffle_batch.py (reference whisk_interface; G1Point/Scalar = %s)" % _backend.inject(),
        "backend": _backend.inject(), "ell": ELL, "n_blinders": N_BLINDERS, "count": count, "crs_bytes": len(crs_bytes),
        "tracker_bytes": 48 * ELL, "proof_bytes": total, "record_bytes": 4 * 48 * ELL + total,
        "crs_seed": CRS_SEED, "proof_seed_base": PROOF_SEED, "sha256": hashlib.sha256(blob).hexdigest(),
        "tampered": [{"base": b, "name": nm, "edits": e, "accepts": v} for (b, nm, e), v in zip(plan, verdicts)],
    }
    with open(os.path.join(out_dir, "shuffle_batch_ell124.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", len(blob), "bytes;", [(t["name"], t["accepts"]) for t in meta["tampered"]])


if __name__ == "__main__":
    main()
