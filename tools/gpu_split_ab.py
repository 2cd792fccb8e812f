#!/usr/bin/env python3
"""One MSM as two window shards on two contexts of the same GPU, in flight together, against the plain single call."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N  # noqa: E402
from tools.gpu_sweep import GX, GY  # noqa: E402


def med(f, reps=21):
    w = []
    for _ in range(reps):
        t = time.perf_counter(); f(); w.append((time.perf_counter() - t) * 1e3)
    w = sorted(w[3:])
    return w[len(w) // 2]


a, b = N.Context(0), N.Context(0)
for c in (a, b):
    c.set_param("profile", 0)
nmax = 1 << 22
dk, dp, ds, dg = a.alloc(32 * nmax), a.alloc(96 * nmax), a.alloc(32 * nmax), a.alloc(96)
dg.upload(GX.to_bytes(48, "little") + GY.to_bytes(48, "little"))
a.gen_scalars_device(dk, nmax, 1)
a.batch_mul_device(dg, 1, dk, dp, nmax)
a.gen_scalars_device(ds, nmax, 2)
a.sync()


def single(n):
    return a.msm_device(dp, ds, n, window_c=16)


def split(n, parts=2, delay_us=0.0):
    a.msm_device_begin(dp, ds, n, window_c=16, shard_rank=0, shard_world=parts)
    if delay_us:                                   # stagger: the second shard's sort chain under the first one's accumulation
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e6 < delay_us:
            pass
    b.msm_device_begin(dp, ds, n, window_c=16, shard_rank=1, shard_world=parts)
    ra = a.msm_device_end()
    rb = b.msm_device_end()
    out = ctypes.create_string_buffer(N.POINT_BYTES)
    N.cg1_add(out, ra, rb)
    return out.raw


for logn in (18, 20, 22):
    n = 1 << logn
    assert N.cg1_eq(single(n), split(n))
    res = {"single": [], "split": []}
    for r in range(3):
        res["single"].append(med(lambda: single(n)))
        res["split"].append(med(lambda: split(n)))
    print(f"2^{logn}: single {min(res['single']):.3f} ms   two window shards in flight {min(res['split']):.3f} ms", flush=True)
    for d in (150, 300, 450, 600, 900):
        print(f"      second shard begun {d} us later: {med(lambda: split(n, delay_us=d)):.3f} ms", flush=True)
