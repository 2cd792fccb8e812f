#!/usr/bin/env python3
"""Throughput of batched Whisk tracker-opening verification (golden fixture cycled)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curdleproofs_pie_amd.shuffle_verifier import OpeningBatchVerifier
g = json.load(open(os.path.join(ROOT, "tests", "golden", "opening_vectors.json")))
items = [((bytes.fromhex(c["r_G"]), bytes.fromhex(c["k_r_G"])), bytes.fromhex(c["k_commitment"]), bytes.fromhex(c["proof"])) for c in g["cases"]]
v = OpeningBatchVerifier()
for n in (1024, 16384, 131072):
    batch = [items[i % len(items)] for i in range(n)]
    assert all(v.verify_many(batch))
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); ok = v.verify_many(batch); ts.append(time.perf_counter() - t0)
        assert all(ok)
    t0 = time.perf_counter(); prep = v.prepare(batch); tp = time.perf_counter() - t0
    print(f"n={n}: {1e3 * min(ts):.1f} ms -> {n / min(ts):.0f} opening proofs/s (host front-end, one thread + Python packing: {1e3 * tp:.1f} ms)", flush=True)
