#!/usr/bin/env python3
"""Randomised differential campaign for the opening-proof verifier (not part of the test-suite): batches of the reference's golden
opening proofs (tests/golden/opening_vectors.json + the torsion cases of torsion_vectors.json) with random damage -- flipped bytes,
infinity / compression / sign flags set or cleared, the response pushed to >= r, points swapped between fields or proofs, points replaced
by a point of order 3 -- go through OpeningBatchVerifier with the front-end on the DEVICE (cg1_opening_prepare_device) and on the HOST;
both must give the same status codes, and both must agree, proof by proof, with cg1_opening_exact: the two equalities the reference
asserts (opening.py:73-74), evaluated one proof at a time on the host without any random weight.

    python tools/gpu_opening_fuzz.py SEED SECONDS
"""
import ctypes
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curdleproofs_pie_amd import _native as N  # noqa: E402
from curdleproofs_pie_amd.shuffle_verifier import OpeningBatchVerifier  # noqa: E402

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
g = json.load(open(os.path.join(ROOT, "tests", "golden", "opening_vectors.json")))
t = json.load(open(os.path.join(ROOT, "tests", "golden", "torsion_vectors.json")))
base = [(bytes.fromhex(c["r_G"]) + bytes.fromhex(c["k_r_G"]), bytes.fromhex(c["k_commitment"]), bytes.fromhex(c["proof"])) for c in g["cases"]]
base += [(bytes.fromhex(c["r_G"]) + bytes.fromhex(c["k_r_G"]), bytes.fromhex(c["k_commitment"]), bytes.fromhex(c["proof"])) for c in t["opening"]
         if len(bytes.fromhex(c["proof"])) == 128]
T3 = bytes.fromhex(t["t3"])
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
rng = random.Random(seed0)
ctx = N.Context(0)
dev, host = OpeningBatchVerifier(ctx, device_front_end=True), OpeningBatchVerifier(ctx, device_front_end=False)
B = 256
t0 = time.time(); it = 0; damaged = 0; accepted = 0; kinds = {}
ok = ctypes.c_int(0)
while time.time() - t0 < budget:
    it += 1
    trk, kcs, pfs = bytearray(), bytearray(), bytearray()
    for s in range(B):
        a, k, p = (bytearray(x) for x in base[rng.randrange(len(base))])
        if rng.random() < 0.5:
            kind = rng.choice(["flip", "flip", "flags", "scalar", "swap", "zero48", "inf", "t3", "steal"])
            kinds[kind] = kinds.get(kind, 0) + 1
            damaged += 1
            fields = [(a, 0), (a, 48), (k, 0), (p, 0), (p, 48)]
            tgt, off = fields[rng.randrange(5)]
            if kind == "flip":
                buf = rng.choice([a, k, p])
                for _ in range(rng.choice([1, 1, 2, 5])):
                    buf[rng.randrange(len(buf))] ^= 1 << rng.randrange(8)
            elif kind == "flags":
                tgt[off] ^= rng.choice([0x80, 0x40, 0x20, 0xC0, 0xE0])
            elif kind == "scalar":
                p[96:128] = rng.choice([R, R + 1, (1 << 256) - 1, R - 1, 0, 1]).to_bytes(32, "little")
            elif kind == "swap":
                t2, o2 = fields[rng.randrange(5)]
                x, y = bytes(tgt[off: off + 48]), bytes(t2[o2: o2 + 48])
                tgt[off: off + 48], t2[o2: o2 + 48] = y, x
            elif kind == "zero48":
                tgt[off: off + 48] = bytes(48)
            elif kind == "inf":
                tgt[off: off + 48] = bytes([0xC0 | rng.choice([0, 0, 0x20])]) + bytes(rng.choice([0, 0, 7]) for _ in range(47))
            elif kind == "t3":
                tgt[off: off + 48] = T3
            elif kind == "steal":                                # a valid point of another proof
                o = base[rng.randrange(len(base))]
                tgt[off: off + 48] = rng.choice([o[0][:48], o[0][48:], o[1], o[2][:48], o[2][48:96]])
        trk += a; kcs += k; pfs += p
    trk, kcs, pfs = bytes(trk), bytes(kcs), bytes(pfs)
    seed = rng.getrandbits(256).to_bytes(32, "little")
    vd = dev.verify_packed(trk, kcs, pfs, _seed=seed); sd = list(dev.last_status)
    vh = host.verify_packed(trk, kcs, pfs, _seed=seed); sh = list(host.last_status)
    if sd != sh:
        print("FRONT-ENDS DIFFER", seed0, it, [(i, x, y) for i, (x, y) in enumerate(zip(sd, sh)) if x != y][:8]); sys.exit(1)
    for i in range(B):
        ctx.check(N.cg1_opening_exact(trk[96 * i: 96 * i + 96], kcs[48 * i: 48 * i + 48], pfs[128 * i: 128 * i + 128], ctypes.byref(ok)))
        if bool(ok.value) != vd[i]:
            print("BATCH VERDICT DIFFERS FROM THE EXACT CHECK", seed0, it, i, sd[i], ok.value,
                  trk[96 * i: 96 * i + 96].hex(), kcs[48 * i: 48 * i + 48].hex(), pfs[128 * i: 128 * i + 128].hex()); sys.exit(1)
    accepted += sum(vd)
print(f"seed {seed0}: {it} batches of {B} opening proofs, {damaged} damaged ({kinds}), {accepted} accepted: device front-end == host front-end == "
      f"the exact per-proof check, every time, in {time.time() - t0:.0f} s")
