import json, os, sys, time
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT)
from curdleproofs_pie_amd import _native as N
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
case = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")))["cases"] if c["ell"] == 124][0]
n = 1024
inst = bytes.fromhex(case["pre_r"] + case["pre_k"] + case["post_r"] + case["post_k"]) * n
proofs = bytes.fromhex(case["proof"]) * n
ctx = N.default_context()
for rep in range(3):
    for big in (False, True):
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), ctx)
        v.prefetch_big = big
        list(v.verify_stream([(inst, proofs, n)] * 3))
        K = 40
        t0 = time.perf_counter()
        for st in v.verify_stream(((inst, proofs, n) for _ in range(K))):
            assert not any(st)
        dt = time.perf_counter() - t0
        print(f"prefetch_big={big}: {1e3*dt/K:.2f} ms per batch -> {n*K/dt:.0f} proofs/s", flush=True)
