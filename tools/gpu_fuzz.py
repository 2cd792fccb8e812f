#!/usr/bin/env python3
"""Extended randomised differential campaign (not part of the test-suite): GPU MSM vs the C oracle's bucket MSM over many
seeds, sizes up to 2^17, all window widths, shard splits and input mixes (few distinct points, structured scalars)."""
import ctypes, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curdleproofs_pie_amd import _native as N
from oracle import bls12_381 as O
from oracle import c_oracle as C

def raw96(p):
    return bytes(96) if p is None else p[0].to_bytes(48, "little") + p[1].to_bytes(48, "little")

def compress_blob(b):
    o = ctypes.create_string_buffer(48); N.cg1_compress(o, b); return o.raw

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
ctx = N.Context(0)
rng = random.Random(seed0)
base = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(64)]
base += [O.g1_neg(p) for p in base[:16]] + [None]
raws = [raw96(p) for p in base]
t0 = time.time(); it = 0
while time.time() - t0 < budget:
    it += 1
    r = rng.random()
    n = rng.randint(1, 200) if r < 0.3 else rng.randint(200, 20000) if r < 0.8 else rng.randint(20000, 140000)
    kind = rng.randrange(7)
    npts = {1: 2, 5: 1}.get(kind, len(raws))
    idx = [rng.randrange(npts) for _ in range(n)]
    def scalar():
        if kind == 2: return rng.choice([0, 1, O.R - 1, 1 << rng.randrange(255), (1 << rng.randrange(1, 255)) - 1])
        if kind == 3: return rng.randrange(1 << rng.choice([8, 16, 20, 33]))
        if kind == 6: return rng.randrange(1 << 20) << rng.choice([16, 100, 230])
        return 0 if rng.random() < 0.02 else rng.randint(0, O.R - 1)
    sc = [scalar() for _ in range(n)]
    if kind == 4: sc = [sc[0]] * n
    p96 = b"".join(raws[i] for i in idx); s32 = b"".join(s.to_bytes(32, "little") for s in sc)
    want = C.compress(C.msm_bucket(p96, s32, n))
    c = rng.choice([0, 0, 0, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16])
    dp, ds = ctx.alloc(96 * n), ctx.alloc(32 * n); dp.upload(p96); ds.upload(s32)
    if rng.random() < 0.25:
        world = rng.choice([2, 3, 5, 8]); cc = c or rng.choice([8, 13, 16])
        acc = ctypes.create_string_buffer(N.POINT_BYTES); N.cg1_identity(acc)
        for rk in range(world):
            N.cg1_add(acc, acc.raw, ctx.msm_device(dp, ds, n, window_c=cc, shard_rank=rk, shard_world=world))
        got = compress_blob(acc.raw)
    else:
        got = compress_blob(ctx.msm_device(dp, ds, n, window_c=c))
    dp.free(); ds.free()
    if got != want:
        print("MISMATCH", dict(seed=seed0, it=it, n=n, kind=kind, c=c), flush=True); sys.exit(1)
    if it % 50 == 0:
        print(f"{it} cases ok ({time.time() - t0:.0f} s)", flush=True)
print(f"fuzz ok: {it} cases, seed {seed0}")
