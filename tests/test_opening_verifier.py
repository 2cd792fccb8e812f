"""Batched Whisk tracker-opening proofs (OpeningBatchVerifier): goldens from the reference's own prover / verifier
(tests/golden/opening_vectors.json, gen_opening_golden.py).  CPU: the statement the native front-end emits is evaluated by
the CPU oracle; GPU: the product path (decompression + merged MSM, per-proof fallback)."""
import ctypes
import json
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from curdleproofs_pie_amd import _native as N  # noqa: E402
from curdleproofs_pie_amd.shuffle_verifier import OpeningBatchVerifier  # noqa: E402
from oracle import c_oracle  # noqa: E402
from oracle.shuffle_check import decompress_affine  # noqa: E402


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(ROOT, "tests", "golden", "opening_vectors.json")) as f:
        return json.load(f)


def items_of(gold):
    items, want = [], []
    for case in gold["cases"]:
        for var in case["variants"]:
            b = {k: bytes.fromhex(case[k]) for k in ("r_G", "k_r_G", "k_commitment", "proof")}
            b.update({k: bytes.fromhex(v) for k, v in var["edits"].items()})
            items.append(((b["r_G"], b["k_r_G"]), b["k_commitment"], b["proof"]))
            want.append(var["accepts"])
    return items, want


def test_statement_matches_reference_verdicts(gold):
    items, want = items_of(gold)
    v = OpeningBatchVerifier()
    prep = v.prepare(items, rng=random.Random(1))
    from oracle import bls12_381 as O

    g96 = O.GX.to_bytes(48, "little") + O.GY.to_bytes(48, "little")     # the generator, from the oracle's constants
    got = []
    for i in range(prep["n"]):
        if prep["status"][i]:
            got.append(False)
            continue
        pts, ok = decompress_affine(prep["points48"].raw[240 * i: 240 * i + 240], 5)
        if not all(ok):
            got.append(False)
            continue
        total = c_oracle.compute_msm(pts + g96, prep["scalars32"].raw[160 * i: 160 * i + 160] + prep["g_scalars32"].raw[32 * i: 32 * i + 32], 6)
        got.append(total == bytes(96))
    assert got == want


def test_challenge_matches_reference(gold):
    """c = (scalar on k_G) / rho_1 must be the challenge the reference verifier drew."""
    R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    rng = random.Random(5)
    items = [((bytes.fromhex(c["r_G"]), bytes.fromhex(c["k_r_G"])), bytes.fromhex(c["k_commitment"]), bytes.fromhex(c["proof"])) for c in gold["cases"]]
    prep = OpeningBatchVerifier().prepare(items, rng=random.Random(5))
    for i, c in enumerate(gold["cases"]):
        rho1 = rng.randint(1, R - 1)
        rng.randint(1, R - 1)
        k_g_scalar = int.from_bytes(prep["scalars32"].raw[160 * i: 160 * i + 32], "little")
        assert k_g_scalar * pow(rho1, -1, R) % R == int.from_bytes(bytes.fromhex(c["challenge"]), "little")


@pytest.mark.gpu
def test_gpu_verdicts(gold):
    items, want = items_of(gold)
    v = OpeningBatchVerifier()
    assert v.verify_many(items, rng=random.Random(2)) == want                      # mixed batch -> per-proof fallback
    good = [it for it, w in zip(items, want) if w]
    assert v.verify_many(good * 20, rng=random.Random(3)) == [True] * (20 * len(good))   # all valid -> merged MSM only
    assert v.verify_many([]) == []
    from curdleproofs_pie_amd.shuffle_verifier import is_valid_whisk_opening_proof
    assert is_valid_whisk_opening_proof(*items[0]) is True and is_valid_whisk_opening_proof(*items[2]) is False
