#!/usr/bin/env python3
"""CPU seconds burnt per wall second while streaming verification batches (detects spin-waiting GPU threads)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
case = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")))["cases"] if c["ell"] == 124][0]
n = 1024
inst = bytes.fromhex(case["pre_r"] + case["pre_k"] + case["post_r"] + case["post_k"]) * n
proofs = bytes.fromhex(case["proof"]) * n
for blocking in (0, 1, 0, 1):
    v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), N.Context(0))
    try:
        v.ctx.set_param("blocking_sync", blocking); v.ctx_msm.set_param("blocking_sync", blocking)
    except Exception as e:
        print("no blocking_sync param:", e)
    list(v.verify_stream([(inst, proofs, n)] * 3))
    K = 30
    c0, t0 = os.times(), time.perf_counter()
    for st in v.verify_stream(((inst, proofs, n) for _ in range(K))):
        assert not any(st)
    c1, dt = os.times(), time.perf_counter() - t0
    cpu = (c1.user - c0.user) + (c1.system - c0.system)
    print(f"blocking_sync={blocking}: {1e3*dt/K:.2f} ms per batch ({n*K/dt:.0f} proofs/s); CPU {cpu/dt:.1f} cores busy (user {c1.user-c0.user:.2f}s sys {c1.system-c0.system:.2f}s over {dt:.2f}s)", flush=True)
