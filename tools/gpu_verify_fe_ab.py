#!/usr/bin/env python3
"""Same-box A/B of the batch verifier with its front-end on the host (all cores) and on the GPU (k_shuffle_front_end, `fe_lanes`
launches side by side): the distinct-proof fixture, batches of 1024, streamed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from batch_fixture import ShuffleBatch
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

fx = ShuffleBatch()
n = int(os.environ.get("BATCH", "1024"))
inst, proofs, _ = fx.tiled(n)
ctx = N.Context(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
threads = int(os.environ.get("THREADS", "0"))
# CONFIGS = "lanes:cus:prio,..." (front-end launches side by side : compute units of their own : wave priority of the front-end kernel)
spec = os.environ.get("CONFIGS", "5:0:0,5:0:3,8:0:3")
configs = [("host front-end", dict(device_front_end=False))] + [
    (f"device front-end, {f} lanes, {c} CUs of their own, priority {pr}, {os.environ.get('PIPELINES', '1')} pipeline(s)",
     dict(device_front_end=True, fe_lanes=int(f), fe_cus=int(c), fe_prio=int(pr), pipelines=int(os.environ.get("PIPELINES", "1"))))
    for f, c, pr in (x.split(":") for x in spec.split(","))]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
for rnd in range(int(os.environ.get('ROUNDS', '2'))):
    for name, kw in configs:
        v = ShuffleBatchVerifier(fx.crs, ctx, threads=threads, **kw)
        list(v.verify_stream([(inst, proofs, n)] * ((4 + kw.get("fe_lanes", 0)) * kw.get("pipelines", 1))))
        acc = {}
        t0 = time.perf_counter()
        for st in v.verify_stream(((inst, proofs, n) for _ in range(K))):
            assert not any(st)
            for k, x in v.last_stats.items():
                if k.endswith("_s"):
                    acc[k] = acc.get(k, 0.0) + x
        dt = time.perf_counter() - t0
        print(f"{name} (host threads {threads or 'all'}): {1e3*dt/K:.2f} ms per batch = {n*K/dt:.0f} proofs/s | " + " ".join(f"{k}={1e3*x/K:.2f}" for k, x in acc.items()), flush=True)
        v.close()
