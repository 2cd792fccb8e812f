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


def test_exact_check_with_status_codes(gold):
    """cg1_opening_exact_status (what OpeningBatchVerifier runs for a handful of proofs -- IsValidWhiskOpeningProof is a batch of one):
    the reference's verdict on every golden variant, and the batch path's status codes (1 bad scalar, 2 bad point, 6 equality)."""
    items, want = items_of(gold)
    st = ctypes.c_int(0)
    R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    seen = set()
    for ((r_g, kr_g), kc, pf), acc in zip(items, want):
        if not (len(r_g) == len(kr_g) == len(kc) == 48 and len(pf) == 128):
            continue
        assert N.cg1_opening_exact_status(r_g + kr_g, kc, pf, ctypes.byref(st)) == 0
        assert (st.value == 0) == acc
        seen.add(st.value)
    (r_g, kr_g), kc, pf = items[0]
    assert N.cg1_opening_exact_status(r_g + kr_g, kc, pf[:96] + R.to_bytes(32, "little"), ctypes.byref(st)) == 0 and st.value == 1
    assert N.cg1_opening_exact_status(bytes(48) + kr_g, kc, pf, ctypes.byref(st)) == 0 and st.value == 2
    assert N.cg1_opening_exact_status(r_g + kr_g, kc, pf[:96] + (R - 1).to_bytes(32, "little"), ctypes.byref(st)) == 0 and st.value == 6
    assert 0 in seen and 6 in seen
    ok = ctypes.c_int(0)
    assert N.cg1_opening_exact(r_g + kr_g, kc, pf, ctypes.byref(ok)) == 0 and ok.value == 1


def test_seed_weights_are_shake256():
    """cg1_opening_weights_from_seed: rho1 | rho2 of proof i = SHAKE256(seed || le64(i))[:32], 16 bytes each, zero-extended to scalar32"""
    import hashlib

    seed = bytes(range(100, 132))
    out = ctypes.create_string_buffer(64 * 5)
    assert N.cg1_opening_weights_from_seed(seed, 3, 5, out) == 0
    for k in range(5):
        d = hashlib.shake_256(seed + (3 + k).to_bytes(8, "little")).digest(32)
        assert out.raw[64 * k: 64 * k + 64] == d[:16] + bytes(16) + d[16:] + bytes(16)


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


def _awkward_items(gold):
    """the goldens + what a hostile sender can do to the wire bytes: infinity flags with payload bits (decoded as the identity by the
    reference, hashed in canonical form), s >= r, s = r - 1, truncated / over-long fields"""
    items, want = items_of(gold)
    (r_g, kr_g), kc, pf = items[0]
    R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    dirty_inf = bytes([0xC0]) + bytes(46) + bytes([7])
    extra = [((r_g, kr_g), kc, pf[:96] + R.to_bytes(32, "little")),              # s = r: Scalar.from_le_bytes raises
             ((r_g, kr_g), kc, pf[:96] + (R - 1).to_bytes(32, "little")),
             ((r_g, kr_g), kc, pf[:96] + bytes([255] * 32)),
             ((dirty_inf, kr_g), kc, pf), ((r_g, dirty_inf), kc, pf), ((r_g, kr_g), dirty_inf, pf),
             ((r_g, kr_g), kc, dirty_inf + pf[48:]), ((r_g, kr_g), kc, pf[:48] + dirty_inf + pf[96:]),
             ((bytes([0xC0]) + bytes(47), bytes([0xC0]) + bytes(47)), bytes([0xC0]) + bytes(47), bytes([0xC0]) + bytes(47) + bytes([0xC0]) + bytes(47) + bytes(32)),
             ((r_g[:47], kr_g), kc, pf), ((r_g, kr_g), kc + b"\0", pf), ((r_g, kr_g), kc, pf[:127]), ((r_g, kr_g), kc, pf + b"tail"),
             ((bytes(48), kr_g), kc, pf), ((r_g, kr_g), bytes([0x80]) + bytes(47), pf)]
    return items + extra, want


@pytest.mark.gpu
def test_device_front_end_equals_host_front_end(gold):
    """cg1_opening_prepare_device against cg1_opening_prepare + the host-side status handling: the same verdicts, the same status codes,
    and -- under the same weights -- the same scalars byte for byte (incl. the summed generator scalar)."""
    items, want = _awkward_items(gold)
    ctx = N.default_context()
    vd, vh = OpeningBatchVerifier(ctx, device_front_end=True), OpeningBatchVerifier(ctx, device_front_end=False)
    for batch in (items, items * 9, [it for it, w in zip(items, want) if w] * 30, items[:1]):
        got_d = vd.verify_many(batch, rng=random.Random(7))
        got_h = vh.verify_many(batch, rng=random.Random(7))
        assert got_d == got_h and vd.last_status == vh.last_status
    assert vd.verify_many(items)[: len(want)] == want
    for seed in (bytes(32), bytes(range(32))):                 # seed-derived weights: the two front-ends draw the same ones
        assert vd.verify_many(items * 5, _seed=seed) == vh.verify_many(items * 5, _seed=seed) and vd.last_status == vh.last_status
    # byte level
    n, trk, kcs, pfs, pre = vd._pack(items * 3)
    weights = vd._weights(n, random.Random(11))
    d_pts, d_sc = ctx.alloc(96 * (5 * n + 1)), ctx.alloc(32 * (5 * n + 1))
    st = (ctypes.c_int32 * n)()
    ps = ctypes.create_string_buffer(5 * n)
    gs = ctypes.create_string_buffer(32 * n)
    ctx.check(N.cg1_opening_prepare_device(ctx.handle, n, trk, kcs, pfs, weights, None, d_pts.ptr, d_sc.ptr, st, ps, gs))
    prep = vh._prepare_host(n, trk, kcs, pfs, None, weights)
    ctx.check(N.cg1_shuffle_apply_point_status(prep["status"], ps.raw, n, 5, prep["scalars32"], prep["g_scalars32"], 1))
    g_sum = ctypes.create_string_buffer(32)
    ctx.check(N.cg1_shuffle_sum_crs_scalars(prep["g_scalars32"], prep["status"], n, 1, g_sum))
    assert list(st) == [int(prep["status"][i]) for i in range(n)]
    assert gs.raw == prep["g_scalars32"].raw[: 32 * n]
    assert d_sc.download(32 * (5 * n + 1)) == prep["scalars32"].raw[: 160 * n] + g_sum.raw
    # the same with the weights derived from a seed on the device / on the host
    seed = bytes(range(7, 39))
    ctx.check(N.cg1_opening_prepare_device(ctx.handle, n, trk, kcs, pfs, None, seed, d_pts.ptr, d_sc.ptr, st, ps, gs))
    prep2 = vh._prepare_host(n, trk, kcs, pfs, None, vh._weights(n, None, seed))
    ctx.check(N.cg1_shuffle_apply_point_status(prep2["status"], ps.raw, n, 5, prep2["scalars32"], prep2["g_scalars32"], 1))
    ctx.check(N.cg1_shuffle_sum_crs_scalars(prep2["g_scalars32"], prep2["status"], n, 1, g_sum))
    assert gs.raw == prep2["g_scalars32"].raw[: 32 * n]
    assert d_sc.download(32 * (5 * n + 1)) == prep2["scalars32"].raw[: 160 * n] + g_sum.raw
    # the decoded points: what the host path's own decompression call leaves there
    d_wire, d_ref, d_stat = ctx.alloc(240 * n), ctx.alloc(96 * 5 * n), ctx.alloc(5 * n)
    d_wire.upload(prep["points48"].raw[: 240 * n])
    ctx.check(N.cg1_batch_decompress_device(ctx.handle, d_wire.ptr, d_ref.ptr, d_stat.ptr, 5 * n, 1))
    assert d_stat.download(5 * n) == ps.raw
    assert d_pts.download(96 * 5 * n) == d_ref.download(96 * 5 * n)


@pytest.mark.gpu
def test_culprit_search_in_slices(gold, monkeypatch):
    """a failing batch larger than one culprit slice: every bad proof is named, nothing else (slices of 64 here; 32 768 in production)"""
    items, want = items_of(gold)
    monkeypatch.setattr(OpeningBatchVerifier, "CULPRIT_SLICE", 64)
    v = OpeningBatchVerifier()
    batch, expect = items * 13, want * 13
    assert len(batch) > 3 * 64 and False in expect
    assert v.verify_many(batch, rng=random.Random(6)) == expect


@pytest.mark.gpu
def test_verify_packed(gold):
    items, want = items_of(gold)
    shaped = [k for k, ((r, kr), kc, pf) in enumerate(items) if len(r) == len(kr) == len(kc) == 48 and len(pf) == 128]
    items, want = [items[k] for k in shaped], [want[k] for k in shaped]
    assert True in want and False in want
    v = OpeningBatchVerifier()
    trk = b"".join(t[0] + t[1] for t, _, _ in items)
    kcs = b"".join(k for _, k, _ in items)
    pfs = b"".join(p for _, _, p in items)
    assert v.verify_packed(trk, kcs, pfs, rng=random.Random(4)) == want
    assert v.verify_packed(b"", b"", b"") == []
    with pytest.raises(ValueError):
        v.verify_packed(trk, kcs, pfs[:-1])
