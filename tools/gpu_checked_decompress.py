#!/usr/bin/env python3
"""k_batch_decompress WITH the subgroup test (the opening verifier's dominant kernel): ms per launch over the opening fixture's points.

    python tools/gpu_checked_decompress.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curdleproofs_pie_amd import _native as N  # noqa: E402

g = json.load(open(os.path.join(ROOT, "tests", "golden", "opening_vectors.json")))
one = b"".join(bytes.fromhex(c[k]) for c in g["cases"] for k in ("k_commitment", "k_r_G", "r_G")) + b"".join(bytes.fromhex(c["proof"])[:96] for c in g["cases"])
assert len(one) % 48 == 0
ctx = N.Context(0)
peak = max(ctx.probe_mad_rate(2, 100) for _ in range(3))
print(f"same-run multiply peak {peak / 1e12:.2f} T/s")
for n in (5 * 1024, 5 * 16384, 5 * 131072, 5 * 1048576):
    wire = (one * (n * 48 // len(one) + 1))[: 48 * n]
    d_w, d_p, d_s = ctx.alloc(48 * n), ctx.alloc(96 * n), ctx.alloc(n)
    d_w.upload(wire)
    for chk in (0, 1):
        ctx.check(N.cg1_batch_decompress_device(ctx.handle, d_w.ptr, d_p.ptr, d_s.ptr, n, chk))
        assert not any(d_s.download(n))
        best = 1e9
        for _ in range(3):
            ctx.timer_begin()
            ctx.check(N.cg1_batch_decompress_enqueue(ctx.handle, d_w.ptr, d_p.ptr, d_s.ptr, n, chk))
            best = min(best, ctx.timer_end())
        print(f"{n:8d} points, subgroup test {'on ' if chk else 'off'}: {best:8.3f} ms = {best * 1e6 / n:6.2f} ns per point", flush=True)
