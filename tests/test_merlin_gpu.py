"""Batched Merlin transcripts on the GPU (k_merlin_batch, SURVEY 8(f) row 1, HIP half): the reference's known answer
(merlin_transcripts/test_merlin.py:33-41), the op sequences recorded from the reference's pure-Python package
(tests/golden/merlin_vectors.json), the opening-proof challenges the reference verifier drew
(tests/golden/opening_vectors.json), and a per-lane differential run against the host transcript including the final
208-byte states."""
import json
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M(native_lib):
    import curdleproofs_pie_amd.merlin as m

    return m


@pytest.fixture(params=["block program", "byte machine", "one lane at a time"], autouse=True)
def kernel_form(request, native_lib):
    """Every test runs on the three kernels behind cg1_merlin_batch_device."""
    ctx = native_lib.default_context()
    rows, sync = {"block program": (1, 1), "byte machine": (0, 1), "one lane at a time": (0, 0)}[request.param]
    ctx.set_param("merlin_rows", rows)
    ctx.set_param("merlin_sync", sync)
    yield {"block program": 2, "byte machine": 1, "one lane at a time": 0}[request.param]
    ctx.set_param("merlin_rows", 1)
    ctx.set_param("merlin_sync", 1)


def test_merlin_known_answer_on_every_lane(M):
    prog = M.TranscriptProgram.__new__(M.TranscriptProgram)
    # test_merlin.py:33-41 uses a plain MerlinTranscript(b"test protocol")
    M.TranscriptProgram.__init__(prog, b"test protocol")
    prog.append(b"some label", 0, 9)
    c = prog.challenge_bytes(b"challenge", 32)
    outs, _ = prog.run([b"some data"] * 130)
    assert all(o[c: c + 32].hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615" for o in outs)


def test_golden_sequences_from_the_reference_package(M):
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "merlin_vectors.json")))["cases"]
    for case in cases:
        prog = M.TranscriptProgram(bytes.fromhex(case["label"]))
        data, checks = b"", []
        for op in case["ops"]:
            lab = bytes.fromhex(op["label"])
            if op["op"] in ("append", "u64"):
                msg = bytes.fromhex(op["msg"]) if op["op"] == "append" else op["x"].to_bytes(8, "little")
                prog.append(lab, len(data), len(msg))
                data += msg
            elif op["op"] == "challenge":
                checks.append((prog.challenge_bytes(lab, op["n"]), op["n"], op["out"]))
            else:
                checks.append((prog.challenge_scalar(lab), 32, op["out"]))
        outs, _ = prog.run([data] * 3)
        for o in outs:
            for off, n, want in checks:
                assert o[off: off + n].hex() == want


def test_opening_proof_challenges_match_the_reference(M):
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "opening_vectors.json")))
    g1 = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")   # test_curdleproofs.py:179-180
    prog = M.TranscriptProgram(b"whisk_opening_proof")
    for k in range(6):                                     # opening.py:66-69: [k_G, G1, k_r_G, r_G, A, B]
        prog.append(b"tracker_opening_proof", 48 * k, 48)
    c = prog.challenge_scalar(b"tracker_opening_proof_challenge")
    rows = []
    for case in gold["cases"]:
        pr = bytes.fromhex(case["proof"])
        rows.append(bytes.fromhex(case["k_commitment"]) + g1 + bytes.fromhex(case["k_r_G"]) + bytes.fromhex(case["r_G"]) + pr[:96])
    outs, _ = prog.run(rows)
    assert [o[c: c + 32].hex() for o in outs] == [case["challenge"] for case in gold["cases"]]


def test_shuffle_shaped_program_equals_host_transcript_per_lane(M, kernel_form, native_lib):
    """300 operations of the shuffle verifier's shape (48-byte points, 32-byte scalars, rejection-sampled challenges, a
    challenge appended back under another label), 200 lanes with different data: outputs AND final sponge states equal the
    host transcript's, lane by lane."""
    rng = random.Random(5)
    n = 200
    prog = M.TranscriptProgram(b"curdleproofs")
    plan, off = [], 0
    for k in range(300):
        r = rng.random()
        if r < 0.7:
            ln = rng.choice([48, 48, 48, 32, 200])
            prog.append(b"curdleproofs_step1" if ln == 48 else b"ipa_step1", off, ln)
            plan.append(("append", b"curdleproofs_step1" if ln == 48 else b"ipa_step1", off, ln))
            off += ln
        elif r < 0.95:
            lab = rng.choice([b"curdleproofs_vec_a", b"ipa_gamma", b"same_msm_gamma"])
            plan.append(("scalar", lab, prog.challenge_scalar(lab), 32))
        else:
            o = prog.challenge_bytes(b"raw", 17)
            prog.append_output(b"echo", o, 17)
            plan.append(("bytes", b"raw", o, 17))
    rows = [bytes(rng.randrange(256) for _ in range(off)) for _ in range(n)]
    outs, states = prog.run(rows, want_states=True)
    assert native_lib.cg1_merlin_last_kernel(native_lib.default_context().handle) == kernel_form       # (no silent fall-back to another kernel)
    for i in (0, 1, 63, 64, 127, 199):
        t = M.CurdleproofsTranscript(b"curdleproofs")
        for kind, lab, o, ln in plan:
            if kind == "append":
                t.append(lab, rows[i][o: o + ln])
            elif kind == "scalar":
                assert bytes(t.get_and_append_challenge(lab).to_le_bytes()) == outs[i][o: o + 32]
            else:
                got = t.challenge_bytes(lab, ln)
                assert got == outs[i][o: o + ln]
                t.append_message(b"echo", got)
        assert bytes(t.strobe._st.raw[:203]) == states[i][:203]
