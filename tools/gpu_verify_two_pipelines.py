#!/usr/bin/env python3
"""Is there slack on the GPU beside one verifier pipeline?  One ShuffleBatchVerifier stream against TWO running side by side on the same
GPU (own contexts, own threads), front-end on the device so that the host does not decide: aggregate proofs/s."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from batch_fixture import ShuffleBatch
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

fx = ShuffleBatch()
n = 1024
inst, proofs, _ = fx.tiled(n)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for fe in (True, False):
    for pipes in (1, 2, 1, 2):
        vs = [ShuffleBatchVerifier(fx.crs, N.Context(0), device_front_end=fe, fe_lanes=3 if pipes == 1 else 2, threads=0 if fe else max(1, int(N.cg1_shuffle_default_threads()) // pipes))
              for _ in range(pipes)]
        for v in vs:
            list(v.verify_stream([(inst, proofs, n)] * 6))
        def work(v):
            for st in v.verify_stream(((inst, proofs, n) for _ in range(K))):
                assert not any(st)
        ths = [threading.Thread(target=work, args=(v,)) for v in vs]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
        print(f"front-end on the {'device' if fe else 'host'}, {pipes} pipeline(s): {1e3 * dt / (K * pipes):.2f} ms per batch = {n * K * pipes / dt:.0f} proofs/s", flush=True)
        for v in vs:
            v.close()
