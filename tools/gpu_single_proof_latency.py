#!/usr/bin/env python3
"""Latency of verifying a FEW shuffle proofs (the drop-in IsValidWhiskShuffleProof is a batch of one) through ShuffleBatchVerifier,
front-end on the device / on the host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from batch_fixture import ShuffleBatch  # noqa: E402
from curdleproofs_pie_amd import _native as N  # noqa: E402
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier  # noqa: E402
fx = ShuffleBatch()
for name, kw in (("default", {}), ("host front-end", {"device_front_end": False}), ("device front-end", {"device_front_end": True})):
    v = ShuffleBatchVerifier(fx.crs, N.Context(0), **kw)
    line = [f"{name} ({'device' if v.device_front_end else 'host'} FE, {v.pipelines} pipeline(s)):"]
    for n in (1, 4, 16, 64, 256):
        inst, proofs, _ = fx.tiled(n)
        assert not any(v.verify_packed(inst, proofs, n))
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); ok = v.verify_packed(inst, proofs, n); ts.append(time.perf_counter() - t0)
            assert not any(ok)
        line.append(f"n={n}: {1e3 * sorted(ts)[3]:.2f} ms")
    print("  ".join(line), flush=True)
    v.close()
