#!/usr/bin/env python3
"""Randomised differential campaign for the verifier's two front-ends (not part of the test-suite): batches of reference-made
ell = 124 proofs with random damage -- flipped bytes anywhere in a proof or its instance, infinity / compression / sign flags
set or cleared, scalars and weights pushed to >= r, points swapped -- go through ShuffleBatchVerifier with the front-end on the
HOST (whose verdicts the tests pin against the reference) and with the front-end on the DEVICE (block program, two pipelines, and
the byte machine): the per-proof status codes must be identical.

    python tools/gpu_verifier_fuzz.py SEED SECONDS
"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from batch_fixture import ShuffleBatch
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 180.0
fx = ShuffleBatch()
rng = random.Random(seed0)
ctx = N.Context(0)
host = ShuffleBatchVerifier(fx.crs, ctx, device_front_end=False)
dev = ShuffleBatchVerifier(fx.crs, N.Context(0), device_front_end=True)
old = ShuffleBatchVerifier(fx.crs, N.Context(0), device_front_end=True, pipelines=1)
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
ib, pb = fx.inst_bytes, fx.proof_bytes
B = 256
t0 = time.time(); it = 0; damaged = 0; rejected = 0; kinds = {}
while time.time() - t0 < budget:
    it += 1
    inst, prf = bytearray(), bytearray()
    for s in range(B):
        i = rng.randrange(fx.count)
        a, p = bytearray(fx.instances[i]), bytearray(fx.proofs[i])
        r = rng.random()
        if r < 0.45:
            kind = rng.choice(["flip", "flip", "flags", "scalar", "swap", "zero48", "inf"])
            kinds[kind] = kinds.get(kind, 0) + 1
            damaged += 1
            tgt = p if rng.random() < 0.6 else a
            if kind == "flip":
                for _ in range(rng.choice([1, 1, 2, 5])):
                    tgt[rng.randrange(len(tgt))] ^= 1 << rng.randrange(8)
            elif kind == "flags":                              # top three bits of a point encoding
                k = 48 * rng.randrange(len(a) // 48) if tgt is a else 48 * rng.randrange((len(p) - 7 * 32) // 48)
                tgt[k] ^= rng.choice([0x80, 0x40, 0x20, 0xC0, 0xE0])
            elif kind == "scalar":                             # one of the proof's seven Fr fields >= r, or just below it
                npts = (len(p) - 7 * 32) // 48
                # (the fields sit where BufReader.read_fr finds them: offsets are recovered from the layout, any 32-byte window works as damage)
                k = rng.randrange(len(p) - 32)
                v = rng.choice([R, R + 1, (1 << 256) - 1, R - 1, 0])
                p[k: k + 32] = v.to_bytes(32, "little")
            elif kind == "swap":
                n48 = len(a) // 48
                x, y = rng.randrange(n48), rng.randrange(n48)
                a[48 * x: 48 * x + 48], a[48 * y: 48 * y + 48] = a[48 * y: 48 * y + 48], a[48 * x: 48 * x + 48]
            elif kind == "zero48":
                k = 48 * rng.randrange(len(tgt) // 48)
                tgt[k: k + 48] = bytes(48)
            else:                                              # a canonical or a sloppy infinity encoding
                k = 48 * rng.randrange(len(tgt) // 48)
                tgt[k: k + 48] = bytes([0xC0]) + (bytes(47) if rng.random() < 0.5 else bytes(rng.randrange(256) for _ in range(47)))
        inst += a; prf += p
    w = bytearray(host.draw_weights(B, rng))
    if rng.random() < 0.3:                                     # a weight >= r in one slot
        s = rng.randrange(B); k = rng.randrange(12)
        w[(s * 12 + k) * 32: (s * 12 + k + 1) * 32] = rng.choice([R, (1 << 256) - 1]).to_bytes(32, "little")
    mode = rng.choice(["merged", "merged", "independent"])
    want = host.verify_packed(bytes(inst), bytes(prf), B, mode=mode, weights=bytes(w))
    got = dev.verify_packed(bytes(inst), bytes(prf), B, mode=mode, weights=bytes(w))
    old.ctx.set_param("fe_rows", it & 1)                       # the single pipeline alternates between the two kernel forms
    for k in range(old.fe_lanes):
        if old._fe[k] is not None:
            old._fe[k][0].set_param("fe_rows", it & 1)
    got1 = old.verify_packed(bytes(inst), bytes(prf), B, mode=mode, weights=bytes(w))
    rejected += sum(1 for x in want if x)
    if got != want or got1 != want:
        bad = [(i, want[i], got[i], got1[i]) for i in range(B) if got[i] != want[i] or got1[i] != want[i]]
        print("MISMATCH", dict(seed=seed0, it=it, mode=mode, slots=bad[:8]), flush=True); sys.exit(1)
    if it % 10 == 0:
        print(f"{it} batches of {B} ok ({time.time() - t0:.0f} s): {damaged} damaged proofs, {rejected} rejected, kinds {kinds}", flush=True)
print(f"verifier fuzz ok: {it} batches of {B} ({it * B} proofs, {damaged} damaged, {rejected} rejected), seed {seed0}; kinds {kinds}")
for v in (host, dev, old):
    v.close()
