"""Native Merlin transcript against the reference's known answers
(merlin_transcripts/merlin_transcripts/test_merlin.py:5-41), against golden op sequences recorded from the
reference's pure-Python package, and -- where /root/reference is present -- live against that package."""
import json
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def M(native_lib):
    import curdleproofs_pie_amd.merlin as m

    return m


def test_strobe_conformance(M):        # test_merlin.py:5-30
    s = M.Strobe128.new(b"Conformance Test Protocol")
    msg = int(99).to_bytes(1, "big") * 1024
    s.meta_ad(b"ms", False); s.meta_ad(b"g", True); s.ad(msg, False)
    s.meta_ad(b"prf", False)
    prf = s.prf(32, False)
    assert prf.hex() == "b48e645ca17c667fd5206ba57a6a228d72d8e1903814d3f17f622996d7cfefb0"
    s.meta_ad(b"key", False); s.key(prf, False)
    s.meta_ad(b"prf", False)
    assert s.prf(32, False).hex() == "07e45cce8078cee259e3e375bb85d75610e2d1e1201c5f645045a194edd49ff8"
    with pytest.raises(AssertionError):     # strobe.py:91: `more` must continue the same operation
        s.ad(b"x", True)


def test_merlin_kat(M):                # test_merlin.py:33-41
    t = M.MerlinTranscript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def _replay(M, case):
    t = M.CurdleproofsTranscript(bytes.fromhex(case["label"]))
    for op in case["ops"]:
        lab = bytes.fromhex(op["label"])
        if op["op"] == "append":
            t.append(lab, bytes.fromhex(op["msg"]))
        elif op["op"] == "u64":
            t.append_u64(lab, op["x"])
        elif op["op"] == "challenge":
            assert t.challenge_bytes(lab, op["n"]).hex() == op["out"]
        else:
            assert bytes(t.get_and_append_challenge(lab).to_le_bytes()).hex() == op["out"]


def test_golden_sequences(M):
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "merlin_vectors.json")))["cases"]
    assert len(cases) >= 10
    for case in cases:
        _replay(M, case)


def test_append_list_equals_repeated_append(M):
    a, b = M.CurdleproofsTranscript(b"x"), M.CurdleproofsTranscript(b"x")
    items = [bytes([i]) * 48 for i in range(20)]
    a.append_list(b"vec", items)
    for it in items:
        b.append(b"vec", it)
    assert a.challenge_bytes(b"c", 32) == b.challenge_bytes(b"c", 32)
    ragged = [b"a", b"bc", b""]
    a.append_list(b"r", ragged)
    for it in ragged:
        b.append(b"r", it)
    assert a.get_and_append_challenges(b"c", 3) == b.get_and_append_challenges(b"c", 3)


@pytest.mark.skipif(not os.path.exists("/root/reference/merlin_transcripts"), reason="reference tree not present")
def test_live_against_reference_package(M):
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference/merlin_transcripts")
    try:
        from merlin_transcripts import MerlinTranscript as Ref
    finally:
        sys.path.pop(0)
    rng = random.Random(5)
    for _ in range(6):
        label = rng.randbytes(rng.randrange(0, 30))
        ours, ref = M.MerlinTranscript(label), Ref(label)
        for _ in range(10):
            lab = rng.randbytes(rng.randrange(1, 12))
            if rng.random() < 0.6:
                msg = rng.randbytes(rng.choice([0, 7, 48, 166, 333, 1000]))
                ours.append_message(lab, msg); ref.append_message(lab, msg)
            else:
                n = rng.choice([1, 32, 200])
                assert ours.challenge_bytes(lab, n) == bytes(ref.challenge_bytes(lab, n))
