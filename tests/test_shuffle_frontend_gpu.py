"""The shuffle verifier's front-end ON THE DEVICE (csrc/kernels_frontend.h: transcript, grand-product scalar, D, A', inner_prod,
challenge inverses -- one proof per lane) against the host front-end (cg1_shuffle_prepare_inputs, whose challenges are the
reference verifier's, tests/test_shuffle_verifier.py): row-input blocks byte for byte and front-end codes, on every golden proof and
every tampered variant (tests/golden/shuffle_vectors.json: reference prover / verifier over the oracle), the challenges inside
the blocks against the ones the REFERENCE recorded, and the derived scalars of the untampered proof's block (beta^-1, inner_prod, the
challenge inverses, the proof's Fr fields) against the oracle's restatement of the verifier (oracle/shuffle_rows.py)."""
import ctypes
import json
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")) as f:
        return json.load(f)


def both_front_ends(N, ctx, v, items, weights, lanes=0, fe_rows=1):
    crs = v.crs
    n = len(items)
    inst, proofs, _ = v.pack(items)
    L, K = crs.points_per_proof, N.cg1_shuffle_rowin_scalars(crs.handle)
    wire = ctypes.create_string_buffer(n * L * 48)
    assert N.cg1_shuffle_gather_points(crs.handle, n, inst, proofs, wire) == 0
    d_wire, d_pts, d_pst = ctx.alloc(n * L * 48), ctx.alloc(n * L * 96), ctx.alloc(n * L)
    d_wire.upload(wire.raw)
    ctx.check(N.cg1_batch_decompress_device(ctx.handle, d_wire.ptr, d_pts.ptr, d_pst.ptr, n * L, 0))
    pts = d_pts.download()
    decoded = b"".join(pts[(i * L + 4 * crs.ell + 1) * 96: (i * L + 4 * crs.ell + 9) * 96] for i in range(n))
    # host
    h_pts, h_rowin, h_status = ctypes.create_string_buffer(n * L * 48), ctypes.create_string_buffer(n * K * 32), (ctypes.c_int32 * n)()
    assert N.cg1_shuffle_prepare_inputs(crs.handle, n, inst, proofs, weights, decoded, 768, h_pts, h_rowin, h_status, 0) == 0
    # device
    aux = ctypes.create_string_buffer(n * 19 * 32)
    assert N.cg1_shuffle_gather_aux(crs.handle, n, proofs, weights, aux) == 0
    d_aux, d_rowin, d_st = ctx.alloc(n * 19 * 32), ctx.alloc(n * K * 32), ctx.alloc(4 * n)
    d_aux.upload(aux.raw)
    fe = N.cg1_shuffle_fe_create(ctx.handle, crs.ell, crs.lg, crs.affine96, crs.bytes)
    assert fe
    assert N.cg1_shuffle_fe_nodes(fe) > 0                     # every golden ell fits the block-program format
    ctx.set_param("fe_rows", fe_rows)                         # 1: block program (k_shuffle_front_end_rows), 0: byte machine (k_shuffle_front_end)
    ctx.check(N.cg1_shuffle_fe_enqueue(fe, ctx.handle, n, d_wire.ptr, d_pts.ptr, d_aux.ptr, d_rowin.ptr, d_st.ptr, lanes))
    ctx.sync()
    assert (N.cg1_shuffle_fe_last_passes(fe, ctx.handle) > 0) == bool(fe_rows)
    ctx.set_param("fe_rows", 1)
    rowin = d_rowin.download()
    st = list((ctypes.c_int32 * n).from_buffer_copy(d_st.download()))
    N.cg1_shuffle_fe_destroy(fe)
    pst = d_pst.download()
    return h_rowin.raw, list(h_status), rowin, st, K, [any(pst[i * L: (i + 1) * L]) for i in range(n)]


@pytest.mark.parametrize("fe_rows", [1, 0])
def test_device_front_end_equals_host_front_end(native_lib, gold, fe_rows):
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
    from test_shuffle_verifier import apply_edits

    N = native_lib
    ctx = N.Context(0)
    for case in gold["cases"]:
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]))
        items = [apply_edits(case, [])] + [apply_edits(case, x["edits"]) for x in case["variants"]]
        items = items * 3                                     # > one wave: lanes of a wave drift apart by rejected draws
        w = v.draw_weights(len(items), random.Random(case["seed"]))
        h_rowin, h_st, d_rowin, d_st, K, bad_pt = both_front_ends(N, ctx, v, items, w, fe_rows=fe_rows)
        assert d_st == h_st, case["ell"]
        assert any(s == 0 for s in h_st)
        for i, s in enumerate(h_st):
            if s == 0 and not bad_pt[i]:
                assert d_rowin[i * K * 32: (i + 1) * K * 32] == h_rowin[i * K * 32: (i + 1) * K * 32], (case["ell"], i)
        # the untampered proof's challenges are the ones the REFERENCE verifier drew (recorded by gen_shuffle_golden.py)
        ell, lg = v.crs.ell, v.crs.lg
        blk = d_rowin[: K * 32]
        slot = lambda k: blk[32 * k: 32 * k + 32].hex()
        ref = case["challenges"]
        got = [slot(8 + 2 * lg + i) for i in range(ell)] + [slot(0), slot(1), slot(2), slot(3), slot(4), slot(5)] + \
              [slot(8 + j) for j in range(lg)] + [slot(6), slot(7)] + [slot(8 + lg + j) for j in range(lg)]
        assert got == [c[1] for c in ref], case["ell"]
        # ... and everything else in that block is what the ORACLE's restatement of the verifier computes from those challenges
        # (oracle/shuffle_rows.py: beta^-1 grand_prod.py:148, inner_prod grand_prod.py:164-166, the round challenges' inverses
        # ipa.py:178 / same_msm.py:175, the proof's Fr fields, the weights) -- not only what the host front-end wrote
        from oracle import shuffle_rows as SR

        Rin = lambda name: {"head": 0, "gam": 8, "gm": 8 + lg, "a": 8 + 2 * lg, "beta_inv": 8 + 2 * lg + ell, "inner_prod": 9 + 2 * lg + ell,
                            "gam_inv": 10 + 2 * lg + ell, "gm_inv": 10 + 3 * lg + ell, "fields": 10 + 4 * lg + ell, "rho": 16 + 4 * lg + ell}[name]
        assert K == Rin("rho") + 12
        val = lambda k: int.from_bytes(blk[32 * k: 32 * k + 32], "little")
        rho = [int.from_bytes(w[32 * j: 32 * j + 32], "little") for j in range(12)]
        ch = [(lab, int.from_bytes(bytes.fromhex(x), "little")) for lab, x in ref]
        fields = SR.proof_fields(bytes.fromhex(case["proof"]), ell)
        _, _, aux = SR.statement_rows(ell, fields, ch, rho)
        assert val(Rin("beta_inv")) == aux["beta_inv"] and val(Rin("inner_prod")) == aux["inner_prod"]
        for j in range(lg):
            assert val(Rin("gam_inv") + j) == SR.inv(val(Rin("gam") + j)) and val(Rin("gm_inv") + j) == SR.inv(val(Rin("gm") + j))
        assert [val(Rin("fields") + j) for j in range(6)] == [fields[k] for k in ("c_final", "d_final", "z_k", "z_t", "z_u", "x_final")]
        assert [val(Rin("rho") + j) for j in range(12)] == rho
        v.close()


def test_few_transcripts_per_wave_and_bad_inputs(native_lib, gold):
    """The same outputs with 16 and with 1 transcript per wave; non-canonical weights and Fr fields get the host's codes."""
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
    from test_shuffle_verifier import apply_edits

    N = native_lib
    ctx = N.Context(0)
    case = gold["cases"][2]
    v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]))
    items = [apply_edits(case, [])] * 5
    w = bytearray(v.draw_weights(5, random.Random(4)))
    w[1 * 12 * 32 + 3 * 32: 1 * 12 * 32 + 4 * 32] = b"\xff" * 32            # proof 1: weight 3 >= r
    ref = None
    for lanes in (0, 16, 1):
        h_rowin, h_st, d_rowin, d_st, K, _ = both_front_ends(N, ctx, v, items, bytes(w), lanes)
        assert d_st == h_st == [0, 4, 0, 0, 0]
        for i in (0, 2, 3, 4):
            assert d_rowin[i * K * 32: (i + 1) * K * 32] == h_rowin[i * K * 32: (i + 1) * K * 32]
        ref = ref or d_rowin
        assert d_rowin == ref
    v.close()
