#!/usr/bin/env python3
"""Throughput of the batch shuffle verifier (BASELINE config 3: ell=124+4 blinders = 128, batches of proofs).
Proofs: the golden ell=124 fixtures cycled (no reference prover on the GPU box); every slot gets its own weights."""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
from test_shuffle_verifier import apply_edits

gold = json.load(open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")))
case = gold["cases"][4]
ctx = N.default_context()
print("host cores:", os.cpu_count(), flush=True)
for threads in (1, 0):
    v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), ctx, threads=threads)
    item = apply_edits(case, [])
    for n in (64, 1024, 4096):
        inst, proofs, _ = v.pack([item] * n)
        for mode in ("merged", "independent"):
            for api in ("packed", "objects"):
                best = None
                for rep in range(3):
                    t0 = time.perf_counter()
                    if api == "packed":
                        ok = [s == 0 for s in v.verify_packed(inst, proofs, n, mode=mode)]
                    else:
                        ok = v.verify_many([item] * n, mode=mode)
                    dt = time.perf_counter() - t0
                    assert all(ok)
                    if best is None or dt < best[0]:
                        best = (dt, dict(v.last_stats))
                st = best[1]
                print(f"threads={threads or 'all'} n={n} {mode} {api}: {best[0]*1e3:.1f} ms -> {n/best[0]:.0f} proofs/s | " +
                      " ".join(f"{k}={1e3*x:.1f}ms" for k, x in st.items() if k.endswith('_s')), flush=True)
