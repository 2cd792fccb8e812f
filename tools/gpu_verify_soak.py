#!/usr/bin/env python3
"""Soak: thousands of verification batches through one verifier stream; checks verdicts, RSS growth and throughput drift."""
import json, os, resource, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
from test_shuffle_verifier import apply_edits
case = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")))["cases"] if c["ell"] == 124][0]
v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), N.Context(0))
good = apply_edits(case, [])
bad = apply_edits(case, [x for x in case["variants"] if x["name"] == "post_r[1] := other point"][0]["edits"])
n = 512
inst_g, proofs_g, _ = v.pack([good] * n)
inst_b, proofs_b, _ = v.pack([good] * 100 + [bad] + [good] * (n - 101))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
def batches():
    for i in range(K):
        yield (inst_b, proofs_b, n) if i % 97 == 5 else (inst_g, proofs_g, n)
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
t0 = time.perf_counter(); t_last = t0
for i, st in enumerate(v.verify_stream(batches())):
    want_bad = (i % 97 == 5)
    assert (st[100] != 0) == want_bad and sum(1 for s in st if s) == (1 if want_bad else 0), (i, [j for j, s in enumerate(st) if s][:5])
    if (i + 1) % 500 == 0:
        now = time.perf_counter()
        print(f"{i + 1} batches ok; last 500: {1e3 * (now - t_last) / 500:.2f} ms per batch; max RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:.0f} MiB", flush=True)
        t_last = now
print(f"soak ok: {K} batches x {n} proofs in {time.perf_counter() - t0:.1f} s; max RSS {rss0 / 1024:.0f} -> {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:.0f} MiB")
