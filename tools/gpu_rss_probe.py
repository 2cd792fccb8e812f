"""Where the batch verifier's resident set comes from: VmRSS / RssAnon / RssFile / RssShmem of the process after each stage (import, contexts,
verifier, first batch, a stream), for `pipelines` = 1 and 3.  (RssShmem / RssFile grow with memory the driver maps into the process:
page-locked staging, and -- on this platform -- device allocations that are host-visible.)

    python tools/gpu_rss_probe.py            -> profiles/r05_rss_probe.txt
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
def rss():
    d = {}
    for line in open("/proc/self/status"):
        k = line.split(":")[0]
        if k in ("VmRSS", "RssAnon", "RssFile", "RssShmem"):
            d[k] = int(line.split()[1]) // 1024
    return d
out = [("start", rss())]
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
out.append(("library loaded", rss()))
ctx = N.Context(0)
out.append(("one context", rss()))
extra = [N.Context(0) for _ in range(4)]
out.append(("five contexts", rss()))
for c in extra: c.close()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
from test_shuffle_verifier import apply_edits
pipelines, n = int(sys.argv[1]), 512
case = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")))["cases"] if c["ell"] == 124][0]
v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), ctx, pipelines=pipelines, device_front_end=True)
out.append(("verifier made", rss()))
good = apply_edits(case, [])
inst, proofs, _ = v.pack([good] * n)
out.append(("one batch packed", rss()))
for st in v.verify_stream([(inst, proofs, n)]):
    assert not any(st)
out.append(("first batch verified", rss()))
for st in v.verify_stream(((inst, proofs, n) for _ in range(60))):
    assert not any(st)
out.append(("60 batches streamed", rss()))
fp = v.footprint()
out.append(("footprint()", {"pinned_mib": fp["pinned_bytes"] >> 20, "device_mib": fp["device_bytes"] >> 20}))
v.close()
out.append(("verifier closed", rss()))
print(json.dumps({"pipelines": pipelines, "stages": out}))
'''


def main():
    for p in (1, 3):
        r = subprocess.run([sys.executable, "-c", CHILD % ROOT, str(p)], capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print("pipelines", p, "failed:", (r.stdout + r.stderr)[-500:])
            continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print("pipelines =", d["pipelines"])
        for name, v in d["stages"]:
            print("   %-24s %s" % (name, v))


if __name__ == "__main__":
    main()
