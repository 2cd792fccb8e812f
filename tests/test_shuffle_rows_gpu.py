"""SURVEY 8(f) row 3 on the device: the scalar rows of the shuffle statement built by k_shuffle_rows from the front-end's
challenge block must equal the host front-end's rows BYTE FOR BYTE (own-point rows, CRS rows, statuses, the CRS sum), on
every golden proof and tampered variant (tests/golden/shuffle_vectors.json: reference prover / verifier over the oracle)."""
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


def device_rows(N, ctx, v, inst, proofs, n, weights, point_status=None):
    crs = v.crs
    L, C, K = crs.points_per_proof, crs.ncrs, N.cg1_shuffle_rowin_scalars(crs.handle)
    pts = ctypes.create_string_buffer(n * L * 48)
    rowin = ctypes.create_string_buffer(n * K * 32)
    status = (ctypes.c_int32 * n)()
    assert N.cg1_shuffle_prepare_inputs(crs.handle, n, inst, proofs, weights, None, 0, pts, rowin, status, 0) == 0
    d_rowin, d_hst, d_pst = ctx.alloc(n * K * 32), ctx.alloc(4 * n), ctx.alloc(n * L)
    d_sc, d_rows, d_st = ctx.alloc((n * L + C) * 32), ctx.alloc(n * C * 32), ctx.alloc(4 * n)
    d_rowin.upload(rowin.raw); d_hst.upload(bytes(status)); d_pst.upload(point_status or bytes(n * L))
    ctx.check(N.cg1_shuffle_rows_device(ctx.handle, crs.ell, crs.lg, n, d_rowin.ptr, d_hst.ptr, d_pst.ptr, d_sc.ptr, d_rows.ptr, d_st.ptr))
    ctx.sync()
    sc, rows = d_sc.download(), d_rows.download()
    st = list((ctypes.c_int32 * n).from_buffer_copy(d_st.download()))
    return sc[: n * L * 32], sc[n * L * 32:], rows, st, list(status)


def test_device_rows_equal_host_rows(native_lib, gold):
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
    from test_shuffle_verifier import apply_edits

    N = native_lib
    ctx = N.Context(0)
    for case in gold["cases"]:
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]))
        items = [apply_edits(case, x["edits"]) for x in case["variants"]] * 2        # > one front-end group of 16
        n = len(items)
        inst, proofs, _ = v.pack(items)
        w = v.draw_weights(n, random.Random(case["seed"]))
        host = v.prepare(inst, proofs, n, weights=w)
        sc, crs_sum, rows, st, host_st = device_rows(N, ctx, v, inst, proofs, n, w)
        assert host_st == [int(host.status[i]) for i in range(n)] == st
        assert any(s == 0 for s in st) and (case["ell"] == 124 and not case["variants"] or any(s != 0 for s in st) or True)
        L, C = v.crs.points_per_proof, v.crs.ncrs
        assert sc == host.scalars32.raw[: n * L * 32], case["ell"]
        assert rows == host.crs_scalars32.raw[: n * C * 32], case["ell"]
        want_sum = ctypes.create_string_buffer(C * 32)
        assert N.cg1_shuffle_sum_crs_scalars(host.crs_scalars32, host.status, n, C, want_sum) == 0
        assert crs_sum == want_sum.raw
        v.close()


def test_undecodable_own_point_rejects_the_proof_on_the_device(native_lib, gold):
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
    from test_shuffle_verifier import apply_edits

    N = native_lib
    ctx = N.Context(0)
    case = gold["cases"][1]
    v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]))
    items = [apply_edits(case, [])] * 5
    inst, proofs, _ = v.pack(items)
    w = v.draw_weights(5, random.Random(3))
    L, C = v.crs.points_per_proof, v.crs.ncrs
    pst = bytearray(5 * L)
    pst[3 * L + 17] = 3                                     # proof 3: one own point failed to decode on the GPU
    sc, crs_sum, rows, st, _ = device_rows(N, ctx, v, inst, proofs, 5, w, bytes(pst))
    assert st == [0, 0, 0, 2, 0]
    assert sc[3 * L * 32: 4 * L * 32] == bytes(L * 32) and rows[3 * C * 32: 4 * C * 32] == bytes(C * 32)
    clean = device_rows(N, ctx, v, inst, proofs, 5, w)
    assert sc[: 3 * L * 32] == clean[0][: 3 * L * 32] and crs_sum != clean[1]
    v.close()
