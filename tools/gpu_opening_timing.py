#!/usr/bin/env python3
"""Throughput of batched Whisk tracker-opening verification (golden fixture cycled): the device front-end (cg1_opening_prepare_device)
against the host one, through the item list of the reference's signature and through the packed entry.

    python tools/gpu_opening_timing.py        -> profiles/r04_opening_fe.txt
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curdleproofs_pie_amd import _native as N  # noqa: E402
from curdleproofs_pie_amd.shuffle_verifier import OpeningBatchVerifier  # noqa: E402

g = json.load(open(os.path.join(ROOT, "tests", "golden", "opening_vectors.json")))
items = [((bytes.fromhex(c["r_G"]), bytes.fromhex(c["k_r_G"])), bytes.fromhex(c["k_commitment"]), bytes.fromhex(c["proof"])) for c in g["cases"]]
ctx = N.default_context()
print(f"host threads of the native pool: {N.cg1_shuffle_default_threads()}")
for n in (1024, 16384, 131072, 1048576):
    batch = [items[i % len(items)] for i in range(n)]
    trk = b"".join(t[0] + t[1] for t, _, _ in batch)
    kcs = b"".join(k for _, k, _ in batch)
    pfs = b"".join(p for _, _, p in batch)
    line = [f"n={n}:"]
    for name, dev in (("device front-end", True), ("host front-end", False)):
        v = OpeningBatchVerifier(ctx, device_front_end=dev)
        assert all(v.verify_many(batch))
        t_items, t_packed = [], []
        for _ in range(3):
            t0 = time.perf_counter(); ok = v.verify_many(batch); t_items.append(time.perf_counter() - t0)
            assert all(ok)
            t0 = time.perf_counter(); ok = v.verify_packed(trk, kcs, pfs); t_packed.append(time.perf_counter() - t0)
            assert all(ok)
        line.append(f"{name}: items {1e3 * min(t_items):.2f} ms = {n / min(t_items) / 1e6:.2f} M proofs/s, packed {1e3 * min(t_packed):.2f} ms = {n / min(t_packed) / 1e6:.2f} M proofs/s;")
    print(" ".join(line), flush=True)

# where a batch of 131 072 goes on the device path: the C call (copies up, gather, checked decompression, transcripts, scalars, copies
# back), the merged MSM, and the Python around them
import ctypes  # noqa: E402
import secrets  # noqa: E402

n = 131072
batch = [items[i % len(items)] for i in range(n)]
trk = b"".join(t[0] + t[1] for t, _, _ in batch)
kcs = b"".join(k for _, k, _ in batch)
pfs = b"".join(p for _, _, p in batch)
v = OpeningBatchVerifier(ctx)
v.verify_packed(trk, kcs, pfs)
for _ in range(3):
    d_pts, d_sc = v._buffers(n)
    t0 = time.perf_counter()
    st = (ctypes.c_int32 * n)(); ps = ctypes.create_string_buffer(5 * n); gs = ctypes.create_string_buffer(32 * n)
    t1 = time.perf_counter()
    ctx.check(N.cg1_opening_prepare_device(ctx.handle, n, trk, kcs, pfs, None, secrets.token_bytes(32), d_pts.ptr, d_sc.ptr, st, ps, gs))
    t2 = time.perf_counter()
    status = memoryview(st).cast("B").cast("i").tolist(); pstat = ps.raw; x = N.ERR_NOT_IN_SUBGROUP in pstat; a = any(s == 0 for s in status)
    t3 = time.perf_counter()
    r = ctx.msm_device(d_pts, d_sc, 5 * n + 1)
    t4 = time.perf_counter()
    out = [s == 0 for s in status]
    t5 = time.perf_counter()
    print(f"split of n={n}: buffers {1e3 * (t1 - t0):.2f}  cg1_opening_prepare_device {1e3 * (t2 - t1):.2f}  status lists {1e3 * (t3 - t2):.2f}  "
          f"merged MSM {1e3 * (t4 - t3):.2f}  verdict list {1e3 * (t5 - t4):.2f} ms", flush=True)
