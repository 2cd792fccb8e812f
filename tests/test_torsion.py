"""Points outside the prime-order subgroup (tests/golden/torsion_vectors.json, gen_torsion_golden.py: reference classes
over the CPU oracle).  The reference decodes unchecked and asserts the same-scalar / opening equalities exactly
(same_scalar.py:101-108, opening.py:73-76); the batch verifiers weight equalities randomly, which is blind to an order-3
defect whenever 3 divides the weight -- so every case is run with weights that ARE multiples of 3 and must still come out
as the reference decided."""
import ctypes
import json
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FR = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


@pytest.fixture(scope="module")
def tors():
    t = json.load(open(os.path.join(ROOT, "tests", "golden", "torsion_vectors.json")))
    assert [c["accepts"] for c in t["opening"]] == [True, False, True, True, True, False, True, True, False]
    assert [c["accepts"] for c in t["shuffle"]["cases"]] == [True, False, True, False, True, True]
    assert [c.get("alpha_mod_3") for c in t["shuffle"]["cases"]][4:] == [1, 1]       # torsion on cm_A / cm_B too, not only on cm_T / cm_U
    return t


class MultiplesOfThree:
    """rng stand-in, the adversary's best case: every weight w is = 1 (mod 3).  The proof's own points carry the scalar
    -w = r - w (the equalities are moved to one side), and r = 1 (mod 3), so every such scalar is a MULTIPLE OF 3: an
    order-3 defect in one of those points is invisible to the weighted check."""

    residue = 1

    def __init__(self):
        self.k = 1

    def randint(self, lo, hi):
        self.k += 7
        w = 3 * (0x1234567 * self.k + (1 << 200)) + self.residue
        assert (FR - w) % 3 == (1 - self.residue) % 3
        return w


class NeverMultiplesOfThree(MultiplesOfThree):
    """every weight w = 0 (mod 3), so every own-point scalar r - w = 1 (mod 3): torsion components are fully visible to the
    weighted check, each under its own weight -- components that cancel in the reference's exact equalities do not cancel here."""
    residue = 0


def test_oracle_agrees_the_torsion_point_has_order_three(tors):
    from oracle import bls12_381 as O

    T3 = O.g1_decompress(bytes.fromhex(tors["t3"]))
    assert T3 == (0, 2) and O.g1_is_on_curve(T3) and not O.g1_in_subgroup(T3)
    assert O.g1_add(O.g1_add(T3, T3), T3) is None


def test_exact_host_checks_match_reference(native_lib, tors):
    """cg1_opening_exact / cg1_shuffle_exact_same_scalar (host, unweighted) == the reference's verdicts."""
    N = native_lib
    ok = ctypes.c_int(-1)
    for c in tors["opening"]:
        b = {k: bytes.fromhex(c[k]) for k in ("r_G", "k_r_G", "k_commitment", "proof")}
        assert N.cg1_opening_exact(b["r_G"] + b["k_r_G"], b["k_commitment"], b["proof"], ctypes.byref(ok)) == 0
        assert bool(ok.value) == c["accepts"], c["name"]
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleCrs

    crs = ShuffleCrs(bytes.fromhex(tors["shuffle"]["crs"]))
    for c in tors["shuffle"]["cases"]:
        inst = bytes.fromhex(c["pre_r"] + c["pre_k"] + c["post_r"] + c["post_k"])
        assert N.cg1_shuffle_exact_same_scalar(crs.handle, inst, bytes.fromhex(c["proof"]), ctypes.byref(ok)) == 0
        assert bool(ok.value) == c["accepts"], c["name"]


@pytest.mark.gpu
def test_opening_batch_with_torsion_points(native_lib, tors):
    from curdleproofs_pie_amd.shuffle_verifier import OpeningBatchVerifier

    items = [((bytes.fromhex(c["r_G"]), bytes.fromhex(c["k_r_G"])), bytes.fromhex(c["k_commitment"]), bytes.fromhex(c["proof"]))
             for c in tors["opening"]]
    want = [c["accepts"] for c in tors["opening"]]
    v = OpeningBatchVerifier(native_lib.Context(0))
    assert v.verify_many(items, rng=MultiplesOfThree()) == want
    assert v.verify_many(items * 40) == want * 40                     # OS-random weights, larger batch
    assert v.verify_many(items[1:2], rng=MultiplesOfThree()) == [False]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["merged", "independent"])
def test_shuffle_batch_with_torsion_points(native_lib, tors, mode):
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    sh = tors["shuffle"]
    v = ShuffleBatchVerifier(bytes.fromhex(sh["crs"]), native_lib.Context(0))
    tr = lambda r, k: [(r[i: i + 48], k[i: i + 48]) for i in range(0, len(r), 48)]
    items = [(tr(bytes.fromhex(c["pre_r"]), bytes.fromhex(c["pre_k"])), tr(bytes.fromhex(c["post_r"]), bytes.fromhex(c["post_k"])),
              bytes.fromhex(c["proof"])) for c in sh["cases"]]
    want = [c["accepts"] for c in sh["cases"]]
    from oracle.shuffle_check import oracle_verdicts

    assert v.verify_many(items, mode=mode, rng=MultiplesOfThree()) == want
    assert v.last_stats["exact_checks"] == 4                          # the four proofs with a point outside G1, nobody else
    # the reference ACCEPTS the proofs whose torsion components cancel, every time: so must we, whatever the weights are
    assert v.verify_many(items, mode=mode, rng=NeverMultiplesOfThree()) == want
    for seed in range(3):
        assert v.verify_many(items, mode=mode, rng=random.Random(seed)) == want
    assert v.verify_many(items * 16, mode=mode) == want * 16
    assert v.verify_many(items[4:] * 3, mode=mode) == [True] * 6 and v.last_stats["exact_checks"] == 6
    assert v.verify_many([items[0], items[2]] * 8, mode=mode, rng=MultiplesOfThree()) == [True] * 16 and v.last_stats["exact_checks"] == 0
    # why flagged proofs are decided apart from the batch -- the weighted statement ALONE (CPU oracle over the same rows)
    # disagrees with the reference in BOTH directions, and differently from one draw of weights to the next:
    #   weights = 1 (mod 3): accepts the two proofs the reference rejects, rejects one the reference accepts;
    #   weights = 0 (mod 3): rejects a cancelling proof the reference accepts;
    #   same-scalar weights w1..w4 zeroed (what _decide_flagged feeds the MSM): all six pass -- the exact host check then
    #   decides the four equalities, as the reference does.
    packed = v.pack(items)[:2]
    assert oracle_verdicts(v, v.prepare(*packed, len(items), rng=MultiplesOfThree())) == [True, True, True, True, False, True]
    assert oracle_verdicts(v, v.prepare(*packed, len(items), rng=NeverMultiplesOfThree())) == [True, False, True, False, False, True]
    w = bytearray(v.draw_weights(len(items), random.Random(1)))
    for i in range(len(items)):
        w[(12 * i + 8) * 32: (12 * i + 12) * 32] = bytes(128)
    assert oracle_verdicts(v, v.prepare(*packed, len(items), weights=bytes(w))) == [True] * 6
    v.close()
