#!/usr/bin/env python3
"""A few whole MSMs of 2^LOGN terms for `rocprofv3 --kernel-trace` (timeline of one call: tools/trace_timeline.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N  # noqa: E402
from tools.gpu_sweep import GX, GY  # noqa: E402
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
kw = {}
if len(sys.argv) > 2:
    kw = {"window_c": 16, "shard_rank": 0, "shard_world": int(sys.argv[2])}
ctx = N.Context(0)
ctx.set_param("profile", 0)
for kv in filter(None, os.environ.get("PARAMS", "").split(",")):          # e.g. PARAMS=rowcol_quad=0,chunk_len=8
    k, v = kv.split("=")
    ctx.set_param(k, int(v))
n = 1 << logn
dk, dp, ds, dg = ctx.alloc(32 * n), ctx.alloc(96 * n), ctx.alloc(32 * n), ctx.alloc(96)
dg.upload(GX.to_bytes(48, "little") + GY.to_bytes(48, "little"))
ctx.gen_scalars_device(dk, n, 1)
ctx.batch_mul_device(dg, 1, dk, dp, n)
ctx.gen_scalars_device(ds, n, 2)
for _ in range(8):
    ctx.msm_device(dp, ds, n, **kw)
