#!/usr/bin/env python3
"""Random campaign over cg1_lincomb_batch (the engine behind a flush of deferred G1Point values): random batch shapes -- 1 ... 400
outputs of 0 ... 700 signed terms over a table of points in G1 (with the identity among them) -- through the GPU paths with and
without the endomorphism split ("glv" 0 / 1: the single-launch kernel for <= 64 combinations, the regime-B chain beyond, the batched
scalar-multiplication kernel for one-term outputs) against the host pool, byte for byte (blobs, affine96, compressed).

    python tools/gpu_lincomb_fuzz.py [seconds] [seed]
"""
import ctypes
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N  # noqa: E402

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    ctx = N.Context(0)
    g = ctypes.create_string_buffer(144); N.cg1_generator(g)
    g96 = ctypes.create_string_buffer(96); N.cg1_to_affine96(g96, g.raw)
    m = 300
    ks = b"".join(rng.randrange(1, R).to_bytes(32, "little") for _ in range(m))
    table = bytearray(ctx.batch_mul_add_host(g96.raw, 1, ks, m, None, m))
    table[96 * 7: 96 * 8] = bytes(96)                      # the identity among the bases
    raw = bytes(table)
    t_end = time.time() + budget
    batches, outputs, terms, paths = 0, 0, 0, {}
    while time.time() < t_end:
        kind = rng.randrange(6)
        if kind == 0:
            shapes = [rng.choice([0, 1, 2, 3, 5, 9, 40, 127, 128, 129, 300, 627, 700]) for _ in range(rng.randrange(1, 8))]
        elif kind == 1:
            shapes = [rng.randrange(4, 40) for _ in range(rng.randrange(2, 64))]
        elif kind == 2:
            shapes = [rng.randrange(4, 30) for _ in range(rng.randrange(65, 400))]          # beyond one single-launch batch: regime B
        elif kind == 3:
            shapes = [rng.choice([1, 1, 1, 2]) for _ in range(rng.randrange(100, 400))]      # shaped outputs: s * B, A + s * B
        elif kind == 4:
            shapes = [rng.randrange(500, 701) for _ in range(rng.randrange(1, 5))]
        else:
            shapes = [rng.choice([0, 1, 2, 3, 7, 20, 64, 200]) for _ in range(rng.randrange(1, 120))]
        offsets, tb, sc = [0], [], []
        for k in shapes:
            for _ in range(k):
                tb.append(rng.randrange(m) | (0x80000000 if rng.random() < 0.3 else 0))
                sc.append(rng.choice([0, 1, 2, R - 1, (R + 1) // 2, rng.randrange(R), rng.randrange(R), rng.randrange(R), rng.randrange(1 << 128)]))
            offsets.append(len(tb))
        n_out = len(shapes)
        offs = (ctypes.c_uint32 * (n_out + 1))(*offsets)
        tba = (ctypes.c_uint32 * max(1, len(tb)))(*tb)
        scb = b"".join(s.to_bytes(32, "little") for s in sc) or bytes(32)
        outs = []
        for path, glv in ((1, 0), (2, 0), (2, 1), (0, 1)):
            ob, oa, ok = (ctypes.create_string_buffer(144 * n_out), ctypes.create_string_buffer(96 * n_out), ctypes.create_string_buffer(48 * n_out))
            used = ctypes.c_int(0)
            ctx.set_param("glv", glv)
            ctx.check(N.cg1_lincomb_batch(ctx.handle, raw, m, offs, n_out, tba, scb, path, ob, oa, ok, ctypes.byref(used)))
            paths[(used.value, glv)] = paths.get((used.value, glv), 0) + 1
            outs.append((ob.raw, oa.raw, ok.raw))
        if not (outs[0] == outs[1] == outs[2] == outs[3]):
            bad = [j for j in range(n_out) if len({o[2][48 * j: 48 * j + 48] for o in outs}) > 1]
            print(f"MISMATCH in batch {batches}: shapes={shapes[:20]}... outputs {bad[:10]}", flush=True)
            sys.exit(1)
        batches += 1; outputs += n_out; terms += len(tb)
        if batches % 50 == 0:
            print(f"{batches} batches ok ({outputs} outputs, {terms} terms; (path used, glv) counts {paths})", flush=True)
    ctx.set_param("glv", 0)
    print(f"lincomb fuzz ok: {batches} batches, {outputs} outputs, {terms} terms in {budget:.0f} s, seed {seed}; (path used, glv) counts {paths}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
