"""Host footprint of the batch verifier against its two knobs: `pipelines` (complete sets of contexts, threads and batch slots) and
`max_pinned_bytes` (page-locked staging per pipeline).  Every configuration runs in a FRESH process (max RSS is a high-water mark):
K batches of 512 proofs through verify_stream, throughput and memory reported.

    python tools/gpu_verify_footprint.py [--batches 300]            -> profiles/r05_verify_footprint.txt
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, resource, sys, time
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from curdleproofs_pie_amd import _native as N
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
from test_shuffle_verifier import apply_edits
pipelines, budget, K, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 512
case = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")))["cases"] if c["ell"] == 124][0]
rss_import = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), N.Context(0), pipelines=(pipelines or None), max_pinned_bytes=(budget or None), device_front_end=True)
good = apply_edits(case, [])
inst, proofs, _ = v.pack([good] * n)
def batches(k):
    for i in range(k):
        yield (inst, proofs, n)
for st in v.verify_stream(batches(40)):                     # warm-up: contexts, slots, code objects
    assert not any(st)
t0 = time.perf_counter()
for st in v.verify_stream(batches(K)):
    assert not any(st)
el = time.perf_counter() - t0
fp = v.footprint()
print(json.dumps({"hw_queues": N.hw_queues(), "pipelines": v.pipelines, "coalesce": v.coalesce, "max_pinned_bytes": budget or None, "ms_per_batch": round(el / K * 1e3, 3), "proofs_per_s": round(n * K / el),
                  "max_rss_mib": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024), "rss_after_import_mib": round(rss_import / 1024),
                  "slot_pinned_mib": round(fp["pinned_bytes"] / 2 ** 20, 1), "slot_device_mib": round(fp["device_bytes"] / 2 ** 20, 1),
                  "slots_per_pipeline": [k["slots"] for k in fp["pipelines"]] or [fp["slots"]]}))
v.close()
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=300)
    a = ap.parse_args()
    rows = []
    for queues, pipelines, budget in ((24, 0, 0), (24, 2, 0), (24, 1, 0), (24, 0, 256 << 20), (4, 0, 0), (4, 0, 512 << 20), (4, 0, 256 << 20)):
        env = dict(os.environ, GPU_MAX_HW_QUEUES=str(queues))
        r = subprocess.run([sys.executable, "-c", CHILD % ROOT, str(pipelines), str(budget), str(a.batches)], capture_output=True, text=True, timeout=900, env=env)
        if r.returncode != 0:
            rows.append({"hw_queues": queues, "pipelines": pipelines, "max_pinned_bytes": budget, "error": (r.stdout + r.stderr)[-600:]})
            continue
        rows.append(json.loads(r.stdout.strip().splitlines()[-1]))
        print(json.dumps(rows[-1]), flush=True)
    print(json.dumps({"batch": 512, "batches": a.batches, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
