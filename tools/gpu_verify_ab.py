#!/usr/bin/env python3
"""Same-box A/B of the batch verifier's switches on the distinct-proof fixture (batches of 1024, streamed):
device_rows (rows built by k_shuffle_rows vs by the host front-end)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from batch_fixture import ShuffleBatch
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

fx = ShuffleBatch()
n = 1024
inst, proofs, _ = fx.tiled(n)
ctx = N.Context(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for rnd in range(2):
    for dev in (True, False):
        v = ShuffleBatchVerifier(fx.crs, ctx, device_rows=dev)
        list(v.verify_stream([(inst, proofs, n)] * 3))
        acc = {}
        t0 = time.perf_counter()
        for st in v.verify_stream(((inst, proofs, n) for _ in range(K))):
            assert not any(st)
            for k, x in v.last_stats.items():
                if k.endswith("_s"):
                    acc[k] = acc.get(k, 0.0) + x
        dt = time.perf_counter() - t0
        print(f"device_rows={dev}: {1e3*dt/K:.2f} ms per batch = {n*K/dt:.0f} proofs/s | " + " ".join(f"{k}={1e3*x/K:.2f}" for k, x in acc.items()), flush=True)
        v.close()
