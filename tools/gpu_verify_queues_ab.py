"""Proofs/s of the batch verifier against the hardware queues the process has and the verifier's shape (batch size, front-end launches side
by side, pipelines): which configuration to pick when the application did NOT raise GPU_MAX_HW_QUEUES (the runtime's default is 4).
Every row is a fresh process (the runtime reads the variable once).

    python tools/gpu_verify_queues_ab.py            -> profiles/r05_verify_queues_ab.txt
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
ROOT = %r
sys.path.insert(0, ROOT)
import bench
from curdleproofs_pie_amd import _native as N
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
batch, fe_lanes, pipelines, steps, coalesce = (int(x) for x in sys.argv[1:6])
ctx = N.Context(0)
fx = bench.load_batch_fixture()
v = ShuffleBatchVerifier(fx.crs, ctx, threads=4, device_front_end=True, fe_lanes=(fe_lanes or None), pipelines=(pipelines or None), coalesce=(None if coalesce < 0 else coalesce))
inst, proofs, want = fx.tiled(batch)
for st in v.verify_stream(((inst, proofs, batch) for _ in range(v.pipelines * (v.fe_lanes + 2) + 2))):
    assert not any(st)
ctx.sync()
t0 = time.perf_counter()
for st in v.verify_stream(((inst, proofs, batch) for _ in range(steps))):
    assert not any(st)
ctx.sync()
el = time.perf_counter() - t0
print(json.dumps({"hw_queues": N.hw_queues(), "batch": batch, "coalesce": v.coalesce, "fe_lanes": v.fe_lanes, "pipelines": v.pipelines, "ms_per_batch": round(el / steps * 1e3, 3),
                  "proofs_per_s": round(batch * steps / el)}))
v.close()
'''


def main():
    rows = []
    import sys as _s
    rows_spec = ((4, 1024, 0, 0, 0), (4, 1024, 0, 0, -1), (4, 2048, 2, 1, 0), (4, 4096, 2, 1, 0), (8, 1024, 0, 0, -1),
                 (24, 1024, 0, 0, 0), (24, 1024, 0, 0, 2048), (24, 1024, 0, 0, 4096), (24, 2048, 0, 0, 0))
    if len(_s.argv) > 1 and _s.argv[1] == "shapes":          # the 24-queue shapes around the default (pipelines x front-end launches x coalescing)
        rows_spec = tuple((24, 1024, fe, pp, co) for pp in (2, 3, 4) for fe in (1, 2, 3) for co in (2048,)) + ((24, 1024, 2, 3, 3072), (24, 1024, 2, 3, 1536), (32, 1024, 2, 4, 2048))
    for queues, batch, fe, pipes, co in rows_spec:
        env = dict(os.environ)
        env["GPU_MAX_HW_QUEUES"] = str(queues)             # said explicitly, 4 included (= the runtime's default): bench.py's import would otherwise raise it
        steps = max(12, 49152 // batch)
        r = subprocess.run([sys.executable, "-c", CHILD % ROOT, str(batch), str(fe), str(pipes), str(steps), str(co)], capture_output=True, text=True, timeout=600, env=env)
        if r.returncode != 0:
            rows.append({"hw_queues": queues, "batch": batch, "error": (r.stdout + r.stderr)[-400:]})
        else:
            rows.append(json.loads(r.stdout.strip().splitlines()[-1]))
        print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
