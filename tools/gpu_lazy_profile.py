"""Where the time of the reference's unchanged control flow goes on the deferred backend: the call-trace replay (tools/replay_call_trace.py)
with stopwatches around the evaluation steps of the Python face.

    python tools/gpu_lazy_profile.py [--reps 5]            -> profiles/r05_lazy_profile.txt
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from curdleproofs_pie_amd import _native as N

    N.tune_runtime()
    import curdleproofs_pie_amd.msm_accumulator as M
    import curdleproofs_pie_amd.py_arkworks_bls12381 as B
    import replay_call_trace as RT

    acc = {}
    depth = [0]

    def timed(mod, name, label=None):
        fn = getattr(mod, name)
        label = label or name

        def w(*x, **kw):
            t0 = time.perf_counter()
            depth[0] += 1
            try:
                return fn(*x, **kw)
            finally:
                depth[0] -= 1
                e = acc.setdefault(label, [0, 0.0])
                e[0] += 1
                e[1] += time.perf_counter() - t0

        setattr(mod, name, w)

    for name in ("_flush", "_decode_leaves", "ensure_normalised", "certify_all", "_certify", "msm_node", "pack_scalars", "_live_after"):
        timed(B, name)
    for name in ("cg1_lincomb_batch", "cg1_batch_decompress_pool", "cg1_batch_subgroup_pool", "cg1_batch_normalize", "cg1_msm"):
        timed(N, name, "native." + name)
    calls = []
    inner = N.cg1_lincomb_batch

    def lincomb(handle, bases, nb, offs, n_out, tba, scb, path, ob, oa, ok, used):
        t0 = time.perf_counter()
        rc = inner(handle, bases, nb, offs, n_out, tba, scb, path, ob, oa, ok, used)
        calls.append((int(n_out), int(offs[n_out]), int(nb), int(used._obj.value), round((time.perf_counter() - t0) * 1e3, 3)))
        return rc

    N.cg1_lincomb_batch = lincomb
    M.ensure_normalised = B.ensure_normalised
    timed(M.MSMAccumulator, "_settle", "acc._settle")
    timed(M.MSMAccumulator, "_final_msm_terms", "acc._final_msm_terms")

    doc, blob = RT.load()
    res = {}
    for rep in range(a.reps):
        rp = RT.product_replayer(doc, blob)
        for ph in ("setup", "prove", "verify"):
            rp.prepare(doc[ph])
        for ph in ("setup", "prove", "verify"):
            acc.clear()
            del calls[:]
            t = rp.run(doc[ph])
            res.setdefault(ph, []).append((t, {k: (v[0], round(v[1] * 1e3, 3)) for k, v in acc.items()}, list(calls)))
        assert not rp.mismatches
    for ph in ("prove", "verify"):
        best = min(res[ph], key=lambda x: x[0])
        print(f"--- {ph}: {best[0] * 1e3:.3f} ms (best of {a.reps}); inclusive times of the evaluation steps (calls, ms)")
        for k, v in sorted(best[1].items(), key=lambda kv: -kv[1][1]):
            print(f"   {k:36s} {v[0]:6d} {v[1]:9.3f}")
        print("   cg1_lincomb_batch calls (outputs, terms, bases, path 1 = pool / 2 = GPU, ms):", " ".join(str(c) for c in best[2]))
    print(json.dumps({"stats": B.stats}))


if __name__ == "__main__":
    main()
