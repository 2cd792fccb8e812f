"""cg1_merlin_batch_device's BLOCK PROGRAM checked without a GPU: `cg1_merlin_block_program_emulate` (test support in the library) runs one transcript through
the node tables the device kernels consume (build_block_program in csrc/capi_frontend.h, walked on the CPU the way k_fill_rows /
k_merlin_batch_rows walk them) -- random operation lists over merlin_transcript.py:11-24 / curdleproofs_transcript.py:15-25 against the
host transcript (itself pinned by the reference's known answers and recorded sequences, tests/test_merlin.py), outputs AND the final
208-byte state.  The kernels run the same tables on the GPU: tests/test_merlin_gpu.py."""
import ctypes
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def M(native_lib):
    import curdleproofs_pie_amd.merlin as m

    return m


def emulate_block_program(prog, row):
    """ONE transcript through the block program on the host (cg1_merlin_block_program_emulate, test support in the library: the node
    tables walked the way the kernels walk them).  -> (output row, 208-byte state, permutations), or None when the operation list
    does not fit the row format (cg1_merlin_batch_device then uses the byte-level kernel)."""
    from curdleproofs_pie_amd import _native as N

    row = bytes(row)
    assert len(row) >= prog.data_bytes
    ops = (N.MerlinOp * max(1, len(prog._ops)))(*prog._ops)
    out = ctypes.create_string_buffer(max(4, prog.out_bytes))
    st = ctypes.create_string_buffer(N.MERLIN_STATE_BYTES)
    passes = ctypes.c_uint32(0)
    rc = N.cg1_merlin_block_program_emulate(prog._init, ops, len(prog._ops), row + b"\0" * 4, len(row), out, len(out), st, ctypes.byref(passes))
    if rc == N.ERR_ARG:
        return None
    assert rc == 0
    return out.raw, st.raw, passes.value


def random_program(M, rng, nops, max_len=300):
    prog = M.TranscriptProgram(bytes(rng.randrange(256) for _ in range(rng.randrange(1, 20))))
    plan, off = [], 0
    labels = [b"", b"a", b"curdleproofs_step1", b"ipa_gamma", bytes(range(32))]
    for _ in range(nops):
        r = rng.random()
        lab = rng.choice(labels)
        if r < 0.55:
            ln = rng.choice([0, 1, 3, 32, 48, 48, 165, 166, 167, rng.randrange(max_len)])
            prog.append(lab, off, ln)
            plan.append(("append", lab, off, ln))
            off += ln
        elif r < 0.8:
            plan.append(("scalar", lab, prog.challenge_scalar(lab), 32))
        elif r < 0.92:
            ln = rng.choice([1, 4, 17, 32, 64, 164])
            plan.append(("bytes", lab, prog.challenge_bytes(lab, ln), ln))
        elif plan and any(p[0] != "append" for p in plan):
            kind, _, o, ln = rng.choice([p for p in plan if p[0] != "append"])
            prog.append_output(lab, o, ln)
            plan.append(("echo", lab, o, ln))
    return prog, plan, off


def host_run_with_prog_label(M, prog, plan, row):
    """The host transcript from the program's own 208-byte initial state (whatever label it was made with)."""
    t = M.CurdleproofsTranscript(b"x")
    ctypes.memmove(t.strobe._st, prog._init, 208)
    out = {}
    for kind, lab, o, ln in plan:
        if kind == "append":
            t.append(lab, row[o: o + ln])
        elif kind == "scalar":
            out[o] = bytes(t.get_and_append_challenge(lab).to_le_bytes())
        elif kind == "bytes":
            out[o] = t.challenge_bytes(lab, ln)
        else:
            t.append_message(lab, out[o][:ln])
    return out, bytes(t.strobe._st.raw[:203])


def test_random_programs_equal_the_host_transcript(M):
    rng = random.Random(11)
    fitted = 0
    for case in range(60):
        prog, plan, nbytes = random_program(M, random.Random(case), rng.randrange(1, 60))
        row = bytes(rng.randrange(256) for _ in range(max(1, nbytes)))
        got = emulate_block_program(prog, row)
        if got is None:                                    # e.g. five self-produced pieces in one block: the byte-level kernel's case
            continue
        fitted += 1
        out, state, passes = got
        want, want_state = host_run_with_prog_label(M, prog, plan, row)
        for o, v in want.items():
            assert out[o: o + len(v)] == v, (case, o)
        assert state[:203] == want_state, case
    assert fitted >= 40


def test_programs_outside_the_row_format_are_refused(M):
    prog = M.TranscriptProgram(b"t")
    prog.challenge_bytes(b"long", 165)                     # a squeeze that would cross the rate
    assert emulate_block_program(prog, b"") is None
    prog = M.TranscriptProgram(b"t")
    c = [prog.challenge_scalar(b"c") for _ in range(6)]
    for o in c:
        prog.append_output(b"", o, 8)                      # six self-produced pieces land in one block (four fit a row)
    assert emulate_block_program(prog, b"") is None


def test_shuffle_shaped_program_on_the_host(M):
    """The 300-operation program of tests/test_merlin_gpu.py, one lane, on the CPU."""
    rng = random.Random(5)
    prog = M.TranscriptProgram(b"curdleproofs")
    plan, off = [], 0
    for k in range(300):
        r = rng.random()
        if r < 0.7:
            ln = rng.choice([48, 48, 48, 32, 200])
            lab = b"curdleproofs_step1" if ln == 48 else b"ipa_step1"
            prog.append(lab, off, ln)
            plan.append(("append", lab, off, ln))
            off += ln
        elif r < 0.95:
            lab = rng.choice([b"curdleproofs_vec_a", b"ipa_gamma", b"same_msm_gamma"])
            plan.append(("scalar", lab, prog.challenge_scalar(lab), 32))
        else:
            o = prog.challenge_bytes(b"raw", 17)
            prog.append_output(b"echo", o, 17)
            plan.append(("bytes", b"raw", o, 17))
            plan.append(("echo", b"echo", o, 17))
    row = bytes(rng.randrange(256) for _ in range(off))
    out, state, passes = emulate_block_program(prog, row)
    want, want_state = host_run_with_prog_label(M, prog, plan, row)
    assert all(out[o: o + len(v)] == v for o, v in want.items()) and state[:203] == want_state
    assert passes > 100
