#!/usr/bin/env python3
"""Verify stream (distinct-proof fixture, batches of 1024) against the size of the native front-end pool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
from batch_fixture import ShuffleBatch
fx = ShuffleBatch()
ctx = N.Context(0)
n = 1024
inst, proofs, want = fx.tiled(n)
print("default threads:", N.cg1_shuffle_default_threads(), "os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
for rep, blocking in ((0, False), (1, True), (2, False), (3, True)):
    print("GPU lanes wait", "asleep (blocking sync)" if blocking else "spinning", flush=True)
    for threads in (8, 12, 14, 16, 18, 20):
        v = ShuffleBatchVerifier(fx.crs, ctx, threads=threads, blocking_sync=blocking)
        list(v.verify_stream([(inst, proofs, n)] * 2))
        K = 16
        acc = {}
        t0 = time.perf_counter()
        for st in v.verify_stream(((inst, proofs, n) for _ in range(K))):
            assert not any(st)
            for k, x in v.last_stats.items():
                if k.endswith("_s"):
                    acc[k] = acc.get(k, 0.0) + x
        dt = time.perf_counter() - t0
        print(f"threads={threads:3d}: {1e3*dt/K:.2f} ms per batch -> {n*K/dt:.0f} proofs/s | " + " ".join(f"{k}={1e3*x/K:.2f}" for k, x in acc.items()), flush=True)
        v.close()
