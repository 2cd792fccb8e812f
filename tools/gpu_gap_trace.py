#!/usr/bin/env python3
"""One MSM size in a loop, for a kernel trace: how much of a call's GPU span is kernels and how much the gaps between dependent launches.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/gpu_gap_trace.py 16
    python tools/gpu_gap_trace.py --parse OUT
"""
import csv
import ctypes
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(logn, reps=30):
    from curdleproofs_pie_amd import _native as N
    ctx = N.Context(0)
    n = 1 << logn
    dk, d_pts, d_sc, dg = ctx.alloc(32 * n), ctx.alloc(96 * n), ctx.alloc(32 * n), ctx.alloc(96)
    g = ctypes.create_string_buffer(N.POINT_BYTES); N.cg1_generator(g)
    a = ctypes.create_string_buffer(96); N.cg1_to_affine96(a, g.raw)
    dg.upload(a.raw)
    ctx.gen_scalars_device(dk, n, 1)
    ctx.batch_mul_device(dg, 1, dk, d_pts, n)
    ctx.gen_scalars_device(d_sc, n, 2)
    for _ in range(reps):
        ctx.msm_device(d_pts, d_sc, n)


def parse(d):
    f = glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    calls, cur = [], []
    for r in rows:
        name = r["Kernel_Name"]
        if "k_batch_mul" in name or "k_gen_scalars" in name:
            continue
        if ("k_prepare" in name) and cur:
            calls.append(cur); cur = []
        cur.append(r)
    calls.append(cur)
    calls = calls[len(calls) // 2:]                          # steady state
    import statistics
    spans, busy, counts = [], [], []
    for c in calls:
        s, e = int(c[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in c)
        spans.append((e - s) / 1e3); busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in c) / 1e3); counts.append(len(c))
    print(f"{len(calls)} calls: {statistics.median(counts)} kernels per call; GPU span first start -> last end {statistics.median(spans):.1f} us, "
          f"sum of kernel durations {statistics.median(busy):.1f} us, gaps {statistics.median(spans) - statistics.median(busy):.1f} us")
    c = calls[-1]
    prev = int(c[0]["Start_Timestamp"])
    for r in c:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  +{(s - prev) / 1e3:6.1f} us gap  {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'][:70]}")
        prev = e


if __name__ == "__main__":
    if sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run(int(sys.argv[1]))
