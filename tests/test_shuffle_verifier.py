"""CPU tests of the batch shuffle-verifier front-end (csrc/shuffle_verify.cpp, curdleproofs_pie_amd/shuffle_verifier.py).

Golden: tests/golden/shuffle_vectors.json -- proofs made by the reference's own prover, with the challenges the
reference's verifier drew and its verdict on every tampered variant (tests/golden/gen_shuffle_golden.py).
The native front-end must (1) reproduce every Fiat-Shamir challenge bit for bit, (2) emit an MSM statement that
evaluates to the identity exactly for the proofs the reference accepts.  The MSM is evaluated here by the CPU oracle
(test infrastructure); the product evaluates it on the GPU (tests/test_shuffle_verifier_gpu.py).
"""
import ctypes
import json
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from curdleproofs_pie_amd import _native as N  # noqa: E402
from curdleproofs_pie_amd.shuffle_verifier import REJECT_LENGTH, ShuffleBatchVerifier, ShuffleCrs  # noqa: E402
from oracle.shuffle_check import decompress_affine, oracle_verdicts  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")
REF = "/root/reference/curdleproofs"


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def trackers(r_hex, k_hex):
    r, k = bytes.fromhex(r_hex), bytes.fromhex(k_hex)
    return [(r[i: i + 48], k[i: i + 48]) for i in range(0, len(r), 48)]


def apply_edits(case, edits):
    bufs = {k: bytearray(bytes.fromhex(case[k])) for k in ("proof", "pre_r", "pre_k", "post_r", "post_k")}
    for which, off, hexbytes in edits:
        b = bytes.fromhex(hexbytes)
        bufs[which][off: off + len(b)] = b
    pre = trackers(bufs["pre_r"].hex(), bufs["pre_k"].hex())
    post = trackers(bufs["post_r"].hex(), bufs["post_k"].hex())
    return pre, post, bytes(bufs["proof"])


def test_layout_sizes(gold):
    for case in gold["cases"]:
        crs = ShuffleCrs(bytes.fromhex(case["crs"]))
        lg = (case["ell"] + 4).bit_length() - 1
        assert crs.ell == case["ell"]
        assert crs.proof_bytes == len(case["proof"]) // 2 == 48 * (19 + 10 * lg) + 32 * 7
        assert crs.points_per_proof == 4 * case["ell"] + 19 + 10 * lg
        assert crs.ncrs == case["ell"] + 9
    with pytest.raises(ValueError):
        ShuffleCrs(bytes.fromhex(gold["cases"][0]["crs"])[:-48])           # ell + 4 not a power of two
    bad = bytearray(bytes.fromhex(gold["cases"][0]["crs"]))
    bad[0] = 0
    with pytest.raises(ValueError):
        ShuffleCrs(bytes(bad))                                              # undecodable CRS point


def test_challenges_match_reference(gold):
    """Every challenge of the reference verifier (recorded in verify order) is reproduced by the native transcript."""
    for case in gold["cases"]:
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]))
        ell, lg = v.crs.ell, v.crs.lg
        inst, proofs, st = v.pack([(trackers(case["pre_r"], case["pre_k"]), trackers(case["post_r"], case["post_k"]), bytes.fromhex(case["proof"]))])
        assert st == [0]
        prep = v.prepare(inst, proofs, 1, rng=random.Random(5), want_challenges=True)
        assert prep.status[0] == 0
        raw = prep.challenges.raw
        got = [raw[32 * i: 32 * i + 32].hex() for i in range(v.crs.challenges_per_proof)]
        head, g_ipa, g_msm, vec_a = got[:8], got[8: 8 + lg], got[8 + lg: 8 + 2 * lg], got[8 + 2 * lg:]
        ref = case["challenges"]
        labels = [c[0] for c in ref]
        vals = [c[1] for c in ref]
        assert labels == (["curdleproofs_vec_a"] * ell + ["same_perm_alpha", "same_perm_beta", "gprod_alpha", "gprod_beta", "ipa_alpha", "ipa_beta"]
                          + ["ipa_gamma"] * lg + ["same_scalar_alpha", "same_msm_alpha"] + ["same_msm_gamma"] * lg)
        assert vec_a == vals[:ell]
        assert head[:6] == vals[ell: ell + 6]
        assert g_ipa == vals[ell + 6: ell + 6 + lg]
        assert head[6:8] == vals[ell + 6 + lg: ell + 8 + lg]
        assert g_msm == vals[ell + 8 + lg:]


def test_golden_verdicts_via_oracle(gold):
    """Statement == identity exactly when the reference accepted (all variants of the small cases; the ell=124 cases
    get the untampered proof and three tampered ones -- the GPU test covers every variant)."""
    for case in gold["cases"]:
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]))
        variants = case["variants"] if case["ell"] <= 28 else [x for x in case["variants"] if x["name"] == "no edit"] + case["variants"][:3]
        items = [apply_edits(case, x["edits"]) for x in variants]
        inst, proofs, st = v.pack(items)
        assert st == [0] * len(items)
        prep = v.prepare(inst, proofs, len(items), rng=random.Random(case["seed"]))
        got = oracle_verdicts(v, prep)
        want = [x["accepts"] for x in variants]
        assert got == want, [(x["name"], g, w) for x, g, w in zip(variants, got, want) if g != w]


def test_weights_do_not_change_verdict_and_scale_statement(gold):
    case = gold["cases"][1]
    v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]))
    item = apply_edits(case, [])
    inst, proofs, _ = v.pack([item])
    a = v.prepare(inst, proofs, 1, rng=random.Random(1))
    b = v.prepare(inst, proofs, 1, rng=random.Random(2))
    assert a.scalars32.raw != b.scalars32.raw
    assert oracle_verdicts(v, a) == oracle_verdicts(v, b) == [True]
    with pytest.raises(AssertionError):
        v.prepare(inst, proofs, 1, weights=bytes(32))
    bad_w = (v.draw_weights(1, random.Random(3)))[:-32] + (2 ** 256 - 1).to_bytes(32, "little")
    assert v.prepare(inst, proofs, 1, weights=bad_w).status[0] == 4


def test_pack_rejects_bad_shapes_and_ignores_trailing_bytes(gold):
    case = gold["cases"][0]
    v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]))
    pre, post, proof = apply_edits(case, [])
    items = [(pre, post, proof), (pre, post, proof + b"\x01\x02"), (pre, post, proof[:-1]), (pre[:-1], post, proof), (pre, post + post[:1], proof)]
    inst, proofs, st = v.pack(items)
    assert st == [0, 0, REJECT_LENGTH, REJECT_LENGTH, REJECT_LENGTH]
    prep = v.prepare(inst, proofs, len(items), rng=random.Random(9))
    assert oracle_verdicts(v, prep)[:2] == [True, True]


def test_threads_agree(gold):
    case = gold["cases"][2]
    items = [apply_edits(case, x["edits"]) for x in case["variants"]]
    w = None
    outs = []
    for threads in (1, 4):
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), threads=threads)
        inst, proofs, _ = v.pack(items)
        w = w or v.draw_weights(len(items), random.Random(3))
        p = v.prepare(inst, proofs, len(items), weights=w)
        outs.append((p.points48.raw, p.scalars32.raw, p.crs_scalars32.raw, list(p.status)))
    assert outs[0] == outs[1]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_differential_against_reference_verifier():
    """Fresh seeded proofs (ell = 12) from the reference prover; EVERY field of the proof is tampered in turn and the
    reference verifier's verdict (IsValidWhiskShuffleProof over our host backend) is compared with the statement's."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_shuffle_golden as G                      # injects the backend, imports the reference

    rng = random.Random(2024)
    random.seed(77)
    ell = 12
    crs = G.CurdleproofsCrs.new(ell, 4)
    pre = G.make_trackers(ell)
    post, proof = G.GenerateWhiskShuffleProof(crs, pre)
    proof = bytes(proof)
    offs, total = G.proof_offsets(4)
    assert total == len(proof)
    other = bytes(G.point_projective_to_bytes(G.G1 * G.Scalar(12345)))
    pre_r, pre_k = G.cat(pre)
    post_r, post_k = G.cat(post)
    items, names = [], []

    def push(name, pr=pre_r, pk=pre_k, qr=post_r, qk=post_k, pf=proof):
        items.append((G.split(pr, pk), G.split(qr, qk), pf))
        names.append(name)

    push("valid")
    push("trailing bytes", pf=proof + b"\x00" * 7)
    for nm, (off, size) in offs.items():
        if size == 48:
            push(nm + " := other", pf=proof[:off] + other + proof[off + 48:])
        else:
            val = (int.from_bytes(proof[off: off + 32], "little") + rng.randint(1, 1000)) % G.FR_MODULUS
            push(nm + " shifted", pf=proof[:off] + val.to_bytes(32, "little") + proof[off + 32:])
    for which in range(4):
        for idx in (0, ell - 1):
            bufs = [bytearray(pre_r), bytearray(pre_k), bytearray(post_r), bytearray(post_k)]
            bufs[which][48 * idx: 48 * idx + 48] = other
            push(f"tracker buf {which}[{idx}] := other", *(bytes(b) for b in bufs))
    want = [bool(G.IsValidWhiskShuffleProof(crs, a, b, c)) for a, b, c in items]
    assert want[0] and want[1] and not any(want[2:])
    v = ShuffleBatchVerifier(crs)
    inst, proofs, st = v.pack(items)
    prep = v.prepare(inst, proofs, len(items), rng=rng)
    got = oracle_verdicts(v, prep)
    assert got == want, [(n, g, w) for n, g, w in zip(names, got, want) if g != w]


def test_decoded_window_path_matches_host_decode(gold):
    """prepare() fed with pre-decoded A/T_1/U_1/B (what the GPU flow does) == prepare() decoding them on the host;
    cg1_shuffle_gather_points == the point layout prepare() emits."""
    case = gold["cases"][3]
    v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]))
    items = [apply_edits(case, x["edits"]) for x in case["variants"]]
    n = len(items)
    inst, proofs, _ = v.pack(items)
    w = v.draw_weights(n, random.Random(4))
    a = v.prepare(inst, proofs, n, weights=w)
    L = v.crs.points_per_proof
    wire = ctypes.create_string_buffer(n * L * 48)
    assert N.cg1_shuffle_gather_points(v.crs.handle, n, inst, proofs, wire) == 0
    assert wire.raw == a.points48.raw
    decoded = b""
    for i in range(n):
        lo = (i * L + 4 * v.crs.ell + 1) * 48
        aff, _ok = decompress_affine(wire.raw[lo: lo + 8 * 48], 8)
        decoded += aff
    b = v.prepare(inst, proofs, n, weights=w, decoded=decoded)
    for i in range(n):
        if a.status[i] == 0:
            assert b.status[i] == 0
            assert a.scalars32.raw[i * L * 32: (i + 1) * L * 32] == b.scalars32.raw[i * L * 32: (i + 1) * L * 32]
        else:
            # an undecodable A/T_1/U_1/B shows up as an all-zero record on the decoded path; the GPU's point status rejects it
            assert a.status[i] in (1, 2, 3)
    assert a.crs_scalars32.raw == b.crs_scalars32.raw or any(a.status[i] == 2 for i in range(n))


def test_grouped_front_end_matches_single(gold):
    """16 transcripts in step with batched (x8, AVX-512 when present) Keccak == one transcript at a time, byte for byte,
    for group sizes that do not divide evenly and for batches mixing valid, tampered and early-rejected proofs."""
    for case_idx, n in ((0, 1), (1, 7), (1, 17), (2, 33), (4, 3)):
        case = gold["cases"][case_idx]
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), threads=2)
        var = case["variants"]
        items = [apply_edits(case, var[i % len(var)]["edits"]) for i in range(n)]
        inst, proofs, _ = v.pack(items)
        w = v.draw_weights(n, random.Random(n))
        outs = []
        for grouped in (0, 1):
            N.cg1_shuffle_set_grouped(grouped)
            try:
                p = v.prepare(inst, proofs, n, weights=w, want_challenges=True)
            finally:
                N.cg1_shuffle_set_grouped(1)
            ok = [i for i in range(n) if p.status[i] == 0]
            C = v.crs.challenges_per_proof * 32
            outs.append((list(p.status), p.points48.raw, p.scalars32.raw, p.crs_scalars32.raw, [p.challenges.raw[i * C: (i + 1) * C] for i in ok]))
        assert outs[0] == outs[1]


def test_keccak_x8_matches_single():
    rng = random.Random(3)
    states = [bytes(rng.getrandbits(8) for _ in range(200)) for _ in range(8)]
    want = []
    for s in states:
        b = ctypes.create_string_buffer(s, 200)
        N.cg1_keccak_f1600(b)
        want.append(b.raw)
    lanes = (ctypes.c_uint64 * 200)()
    for k, s in enumerate(states):
        for w in range(25):
            lanes[8 * w + k] = int.from_bytes(s[8 * w: 8 * w + 8], "little")
    N.cg1_keccak_f1600_x8(lanes)
    for k in range(8):
        assert b"".join(int(lanes[8 * w + k]).to_bytes(8, "little") for w in range(25)) == want[k]
    # sponges permuted where they lie (AVX-512: in-register transposes), any number 1..8 of them
    for live in (1, 3, 8):
        bufs = [ctypes.create_string_buffer(s, 200) for s in states[:live]]
        ptrs = (ctypes.c_void_p * live)(*[ctypes.addressof(b) for b in bufs])
        N.cg1_keccak_f1600_x8_states(ptrs, live)
        assert [b.raw for b in bufs] == want[:live]
    # known answer: Keccak-f[1600] of the all-zero state (first lane), keccak.py:16-66
    z = ctypes.create_string_buffer(200)
    N.cg1_keccak_f1600(z)
    assert z.raw[:8].hex() == "e7dde140798f25f1"
