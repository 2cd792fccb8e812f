#!/usr/bin/env python3
"""Same-process A/B of context parameters on whole MSMs: python tools/gpu_param_ab.py LOGN name=v1,v2,... [name2=...]
Every combination is timed in turn (interleaved, 3 passes, median of 9 calls each) with profile 1; phases from one profile-2 call."""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N  # noqa: E402
from tools.gpu_sweep import GX, GY  # noqa: E402

logn = int(sys.argv[1])
axes = []
for a in sys.argv[2:]:
    k, vs = a.split("=")
    axes.append((k, [int(v) for v in vs.split(",")]))
n = 1 << logn
ctx = N.Context(0)
dk, dp, ds, dg = ctx.alloc(32 * n), ctx.alloc(96 * n), ctx.alloc(32 * n), ctx.alloc(96)
dg.upload(GX.to_bytes(48, "little") + GY.to_bytes(48, "little"))
ctx.gen_scalars_device(dk, n, 1); ctx.batch_mul_device(dg, 1, dk, dp, n); ctx.gen_scalars_device(ds, n, 2)
ref = None
combos = list(itertools.product(*[vs for _, vs in axes]))
times = {c: [] for c in combos}
win = int(os.environ.get("WINDOW", "0"))
for rnd in range(3):
    for c in combos:
        for (k, _), v in zip(axes, c):
            ctx.set_param(k, v)
        ctx.set_param("profile", 1)
        r = ctx.msm_device(dp, ds, n, window_c=win)
        if ref is None:
            ref = r
        assert N.cg1_eq(r, ref), ("result differs", c)
        for _ in range(9):
            t = time.perf_counter(); ctx.msm_device(dp, ds, n, window_c=win); times[c].append((time.perf_counter() - t) * 1e3)
for c in combos:
    for (k, _), v in zip(axes, c):
        ctx.set_param(k, v)
    ctx.set_param("profile", 2)
    ctx.msm_device(dp, ds, n, window_c=win)
    tm = ctx.timings()
    s = sorted(times[c])
    print(f"2^{logn} " + " ".join(f"{k}={v}" for (k, _), v in zip(axes, c)) + f": median {s[len(s)//2]:.3f} ms min {s[0]:.3f} | c={tm['window_c']} " +
          " ".join(f"{k}={v:.3f}" for k, v in tm.items() if k in ("prepare", "sort_count", "sort_scatter", "chunks", "accumulate", "seg_reduce", "bit_tree", "host_tail")), flush=True)
