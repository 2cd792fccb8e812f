"""The device front-end's BLOCK PROGRAM (csrc/kernels_frontend.h, second form) checked without a GPU: the host cuts the verifier's
transcript into the rate blocks between two Keccak permutations (build_block_program in csrc/capi_frontend.h) and k_fill_rows /
k_shuffle_front_end_rows consume them.  `cg1_shuffle_fe_emulate_to_first_barrier` walks those very tables on the CPU for one proof
-- rows, late pieces, first-draw and redo squeeze nodes -- up to the grand-product step; the challenges it draws must be the ones
the REFERENCE verifier drew (tests/golden/shuffle_vectors.json: curdleproofs.py:176-180, same_perm.py:91-96 over
merlin_transcripts, recorded by gen_shuffle_golden.py).  The rest of the program (same mechanisms, plus the three compute steps)
is compared on the GPU byte for byte: tests/test_shuffle_frontend_gpu.py."""
import ctypes
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")) as f:
        return json.load(f)


def test_program_shape_for_the_supported_sizes(native_lib):
    N = native_lib
    for lg in range(3, 11):
        ell = (1 << lg) - 4
        out = (ctypes.c_uint32 * 4)()
        assert N.cg1_shuffle_fe_program_shape(ell, lg, out) == 0
        ops, nodes, squeeze, pieces = list(out)
        n_chal = ell + 8 + 2 * lg                                   # curdleproofs_transcript.py:15-25 is called this often per verification
        assert ops == (7 * ell + 10 * lg + 36) + n_chal + 3          # messages appended + challenges + the three compute steps
        assert nodes > 0, ell                                       # the program fits the row format
        assert squeeze == 2 * n_chal                                # a first-draw node and a redo node per challenge
        assert 1 <= pieces <= 5
    out = (ctypes.c_uint32 * 4)()
    assert N.cg1_shuffle_fe_program_shape(124, 7, out) == 0 and list(out)[:3] == [1123, 710, 292]
    assert N.cg1_shuffle_fe_program_shape(124, 6, out) != 0         # ell + 4 must be 2^lg
    assert N.cg1_shuffle_fe_program_shape(0, 2, out) != 0


def test_block_program_draws_the_reference_challenges(native_lib, gold):
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
    from test_shuffle_verifier import trackers

    N = native_lib
    for case in gold["cases"]:
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]))
        crs = v.crs
        ell, lg = crs.ell, crs.lg
        inst, proofs, st = v.pack([(trackers(case["pre_r"], case["pre_k"]), trackers(case["post_r"], case["post_k"]), bytes.fromhex(case["proof"]))])
        assert st == [0]
        L, K = crs.points_per_proof, N.cg1_shuffle_rowin_scalars(crs.handle)
        wire = ctypes.create_string_buffer(L * 48)
        assert N.cg1_shuffle_gather_points(crs.handle, 1, inst, proofs, wire) == 0
        out_row = ctypes.create_string_buffer((K + 6) * 32)
        passes = ctypes.c_uint32(0)
        h48 = crs.bytes[(ell + 4) * 48: (ell + 5) * 48]
        assert N.cg1_shuffle_fe_emulate_to_first_barrier(ell, lg, h48, wire.raw, out_row, len(out_row), ctypes.byref(passes)) == 0
        slot = lambda k: out_row.raw[32 * k: 32 * k + 32].hex()
        ref = [c[1] for c in case["challenges"]]
        assert [slot(8 + 2 * lg + i) for i in range(ell)] == ref[:ell], ell         # curdleproofs_vec_a
        assert [slot(0), slot(1)] == ref[ell: ell + 2], ell                          # same_perm_alpha, same_perm_beta
        assert slot(2) == "00" * 32                                                  # nothing drawn past the barrier
        # permutations up to there: at least one per challenge (the PRF's forced permutation) besides the full blocks of the 4 ell + 1
        # point messages; rejected draws (probability 0.55 each) add redo blocks
        assert passes.value >= (ell + 2) + (4 * ell + 1) * 74 // 166, (ell, passes.value)
        v.close()
