#!/usr/bin/env python3
"""Front-end sweep on the GPU box (mean of 8 steps, not best-of): threads, pipeline chunk, grouped transcripts on/off."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
case = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")))["cases"] if c["ell"] == 124][0]
ctx = N.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
inst = bytes.fromhex(case["pre_r"] + case["pre_k"] + case["post_r"] + case["post_k"]) * n
proofs = bytes.fromhex(case["proof"]) * n
print("default threads:", N.cg1_shuffle_default_threads(), "batch", n, flush=True)
for grouped in (1,):
    N.cg1_shuffle_set_grouped(grouped)
    for threads in (1, 0):
        for chunk in (256, 1 << 20):
            v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), ctx, threads=threads, chunk=chunk)
            v.verify_packed(inst, proofs, n)
            ts = []
            for _ in range(8 if threads != 1 else 2):
                t0 = time.perf_counter(); st = v.verify_packed(inst, proofs, n); ts.append(time.perf_counter() - t0)
            assert not any(st)
            ts.sort()
            print(f"grouped={grouped} threads={threads or 'all'} chunk={chunk}: mean {1e3*sum(ts)/len(ts):.1f} ms  min {1e3*ts[0]:.1f}  max {1e3*ts[-1]:.1f}  -> {n*len(ts)/sum(ts):.0f} proofs/s | "
                  + " ".join(f"{k}={1e3*x:.1f}" for k, x in v.last_stats.items() if k.endswith('_s')), flush=True)
N.cg1_shuffle_set_grouped(1)
for chunk in (256, 512, 1024, 256, 512, 1024):
    v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), ctx, chunk=chunk)
    list(v.verify_stream([(inst, proofs, n)] * 2))
    K = 12
    t0 = time.perf_counter()
    for st in v.verify_stream(((inst, proofs, n) for _ in range(K))):
        assert not any(st)
    dt = time.perf_counter() - t0
    print(f"stream of {K} batches, chunk={chunk}: {1e3*dt/K:.1f} ms per batch -> {n*K/dt:.0f} proofs/s | last: "
          + " ".join(f"{k}={1e3*x:.1f}" for k, x in v.last_stats.items() if k.endswith('_s')), flush=True)
