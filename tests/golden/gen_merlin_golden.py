#!/usr/bin/env python3
"""Generate tests/golden/merlin_vectors.json by IMPORTING the reference's pure-Python Merlin package
(/root/reference/merlin_transcripts -- stdlib only, importable in the build container) and
recording seeded op sequences with their outputs.  Only data (inputs / expected outputs) is committed.

    python tests/golden/gen_merlin_golden.py
"""
import json
import os
import random
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/merlin_transcripts")
from merlin_transcripts import MerlinTranscript  # noqa: E402

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def main():
    rng = random.Random(20241008)
    cases = []
    for ci in range(12):
        label = bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 5, 12, 40])))
        t = MerlinTranscript(label)
        ops = []
        for _ in range(rng.randrange(3, 14)):
            kind = rng.choice(["append", "append", "u64", "challenge", "scalar"])
            lab = bytes(rng.randrange(256) for _ in range(rng.choice([1, 3, 9, 20])))
            if kind == "append":
                n = rng.choice([0, 1, 31, 32, 48, 165, 166, 167, 332, 500, 2000])    # around the 166-byte rate
                msg = bytes(rng.randrange(256) for _ in range(n))
                t.append_message(lab, msg)
                ops.append({"op": "append", "label": lab.hex(), "msg": msg.hex()})
            elif kind == "u64":
                x = rng.randrange(1 << 64)
                t.append_u64(lab, x)
                ops.append({"op": "u64", "label": lab.hex(), "x": x})
            elif kind == "challenge":
                n = rng.choice([0, 1, 32, 64, 166, 400])
                out = t.challenge_bytes(lab, n)
                ops.append({"op": "challenge", "label": lab.hex(), "n": n, "out": bytes(out).hex()})
            else:  # curdleproofs_transcript.py:15-25 restated on the reference transcript
                while True:
                    cb = bytes(t.challenge_bytes(lab, 32))
                    v = int.from_bytes(cb, "little")
                    if v >= R or v == 0:
                        continue
                    t.append_message(lab, cb)
                    break
                ops.append({"op": "scalar", "label": lab.hex(), "out": cb.hex()})
        cases.append({"label": label.hex(), "ops": ops})
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "merlin_vectors.json")
    json.dump({"generator": "tests/golden/gen_merlin_golden.py (reference merlin_transcripts imported)", "cases": cases}, open(out, "w"))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
