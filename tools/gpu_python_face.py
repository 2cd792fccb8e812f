"""Wall time of the drop-in itself: compute_MSM / MSMAccumulator through the REFERENCE'S signature (lists of G1Point / Scalar
objects, msm_accumulator.py:6-12,37-68), with the host-marshalling / device split.  bench.py's `python_face` object is
`measure()`; run as a script for the longer table (profiles/r04_python_face.txt).

    python tools/gpu_python_face.py [--sizes 16,20] [--reps 5]
"""
import argparse
import ctypes
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

R = 52435875175126190479447740508185965837690552500527637822603658699938581184513


def make_points(n, seed, projective):
    """n distinct G1Point objects k_i G made on the GPU (k_batch_mul); projective=True re-randomises every Z on the host
    ((P + Q) - Q leaves the value and a Z != 1), which is what util.get_random_point() = G * random_scalar() hands the callers."""
    import curdleproofs_pie_amd as A
    from curdleproofs_pie_amd.msm_accumulator import batch_mul

    rng = random.Random(seed)
    ks = [rng.randint(1, R - 1) for _ in range(n)]
    pts = batch_mul([A.G1Point()] * n, [A.Scalar._raw(k) for k in ks])
    if projective:
        import curdleproofs_pie_amd.py_arkworks_bls12381 as B

        prev = B.set_lazy(False)           # computed at once on the host library: the blobs keep Z != 1 (a deferred value would leave its evaluation normalised)
        try:
            Q = A.G1Point() * A.Scalar(12345)
            pts = [(p + Q) - Q for p in pts]
        finally:
            B.set_lazy(prev)
    return pts, ks


def time_compute_msm(n, projective, reps, seed=1, check=True):
    import curdleproofs_pie_amd as A
    import curdleproofs_pie_amd.msm_accumulator as M
    import curdleproofs_pie_amd.py_arkworks_bls12381 as B
    from curdleproofs_pie_amd import _native as N

    note = lambda m: print("[gpu_python_face] n = %d %s: %s" % (n, "projective" if projective else "normal forms", m), file=sys.stderr, flush=True)
    note("making the points")
    pts, ks = make_points(n, seed, projective)
    rng = random.Random(seed + 100)
    sc_ints = [rng.randint(0, R - 1) for _ in range(n)]
    scalars = [A.Scalar._raw(v) for v in sc_ints]
    ctx = N.default_context()
    M.clear_vec_cache()
    res = {"n": n, "inputs": "projective blobs (Z != 1)" if projective else "normal forms (Z = 1)"}
    note("warm-up call")
    A.compute_MSM(pts[:64], scalars[:64])                      # library warm-up (context, staging growth is part of 'first')
    note("first call")
    t0 = time.perf_counter()
    first = A.compute_MSM(pts, scalars)
    res["first_call_ms"] = (time.perf_counter() - t0) * 1e3    # includes growing the page-locked staging
    path0 = M.last_path
    if check:
        tot = sum(k * s for k, s in zip(ks, sc_ints)) % R
        assert first == A.G1Point() * A.Scalar(tot), "compute_MSM differs from its closed form"
    # steady state, fresh lists every time so the identity cache never hits: the "blobs" path
    walls, dev = [], []
    note("steady state (%s)" % path0)
    for _ in range(reps):
        b2 = list(pts)
        b2[0], b2[1] = b2[1], b2[0]
        s2 = list(scalars)
        s2[0], s2[1] = s2[1], s2[0]
        M._vec_seen.clear()
        t0 = time.perf_counter()
        got = A.compute_MSM(b2, s2)
        walls.append((time.perf_counter() - t0) * 1e3)
        assert M.last_path == path0 and got == first
        t = ctx.timings()
        dev.append(t["host_enqueue"] + t["host_wait"] + t["host_events"] + t["host_horner"])
    res["path"] = path0
    if n >= M._SLICED_UPLOAD_MIN:
        res["sliced_upload_ms"] = dict(zip(("scalar_walks", "point_walks", "msm_after_fence"), (round(v, 3) for v in M.last_pack_ms)))
    res["wall_ms"] = min(walls)
    res["wall_ms_median"] = sorted(walls)[len(walls) // 2]
    res["device_call_ms"] = min(dev)                           # enqueue + wait + Horner inside cg1_msm_blobs (H2D queued in front of it)
    # the marshalling alone
    st = M._staging(ctx)
    t0 = time.perf_counter(); B.pack_points(pts, st.points(n), st.cap_pts); t1 = time.perf_counter()
    B.pack_scalars(scalars, st.scalars(n), st.cap_sc); t2 = time.perf_counter()
    res["pack_points_ms"] = (t1 - t0) * 1e3
    res["pack_scalars_ms"] = (t2 - t1) * 1e3
    res["h2d_bytes"] = 176 * n
    note("resident vector")
    # resident vector: the same list object again and again (only scalars move)
    A.compute_MSM(pts, scalars); A.compute_MSM(pts, scalars)
    walls = []
    for _ in range(reps):
        t0 = time.perf_counter()
        got = A.compute_MSM(pts, scalars)
        walls.append((time.perf_counter() - t0) * 1e3)
        assert M.last_path == "resident" and got == first
    res["resident_wall_ms"] = min(walls)
    res["scalar_muls_per_s_seen_by_caller"] = n / (res["wall_ms"] * 1e-3)
    M.clear_vec_cache()
    return res


def time_accumulator(reps=5):
    """The reference's recorded accumulate_check call sequence of one N = 128 verification (tests/golden/accumulator_vectors.json:
    8 calls, 1 013 pairs, recorded rho) replayed through MSMAccumulator + verify()."""
    import curdleproofs_pie_amd as A

    with open(os.path.join(ROOT, "tests", "golden", "accumulator_vectors.json")) as f:
        data = json.load(f)
    seqs = data.get("sequences") or []
    if not seqs:
        return None
    seq = max(seqs, key=lambda s: sum(len(c["bases"]) for c in s["calls"]))      # the N = 128 sequence: 8 calls, 1 013 pairs
    dec = {}

    def pt(h):
        p = dec.get(h)
        if p is None:
            p = dec[h] = A.G1Point.from_compressed_bytes_unchecked(bytes.fromhex(h))
        return p

    calls = [(pt(c["C"]), [pt(b) for b in c["bases"]], [A.Scalar.from_le_bytes(bytes.fromhex(s)) for s in c["scalars"]]) for c in seq["calls"]]
    out = {"calls": len(calls), "pairs": sum(len(c[1]) for c in calls)}
    first_acc, first_ver, acc_ms, ver_ms = None, None, [], []
    for r in range(reps + 1):
        # (from the second replay on the points keep their normal forms: CRS points met before)
        acc = A.MSMAccumulator()
        t0 = time.perf_counter()
        for C, bases, scalars in calls:
            acc.accumulate_check(C, bases, scalars)
        t1 = time.perf_counter()
        try:
            acc.verify()
            ok = True
        except AssertionError:
            ok = False
        t2 = time.perf_counter()
        if r == 0:
            first_acc, first_ver = (t1 - t0) * 1e3, (t2 - t1) * 1e3
        else:
            acc_ms.append((t1 - t0) * 1e3); ver_ms.append((t2 - t1) * 1e3)
    out["verdict"] = ok
    out.update({"first_accumulate_ms": first_acc, "first_verify_ms": first_ver, "accumulate_ms": min(acc_ms), "verify_ms": min(ver_ms),
                "final_msm_terms": len(acc.base_scalar_map) + len(acc._lhs)})
    return out


def measure(sizes=(16, 20), reps=3):
    out = {"what": "wall time of compute_MSM(bases, scalars) over lists of G1Point / Scalar objects (msm_accumulator.py:6-12), fresh lists each call",
           "helper": "csrc/pyface.c" if __import__("curdleproofs_pie_amd.py_arkworks_bls12381", fromlist=["_pyface"])._pyface is not None else "pure Python"}
    for lg in sizes:
        n = 1 << lg
        out["2^%d" % lg] = time_compute_msm(n, True, reps)
        out["2^%d_normal_forms" % lg] = {k: v for k, v in time_compute_msm(n, False, reps, check=False).items() if k in ("wall_ms", "device_call_ms", "path", "resident_wall_ms")}
    try:
        out["accumulator_n128"] = time_accumulator()
    except (OSError, KeyError, ValueError) as e:
        out["accumulator_n128"] = {"skipped": str(e)}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="7,10,12,16,20")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    print(json.dumps(measure(tuple(int(x) for x in a.sizes.split(",")), a.reps), indent=1))
