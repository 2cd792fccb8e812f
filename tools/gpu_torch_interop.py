#!/usr/bin/env python3
"""torch (its own HIP runtime copy) and libcurdle_g1.so in one process; torch tensors' data_ptr() as MSM inputs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from curdleproofs_pie_amd import _native as N  # noqa: E402

x = torch.ones(1024, device="cuda")
print("torch sum", float(x.sum()))
ctx = N.Context(0)
n = 1 << 12
GX = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
GY = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
dk, dp, ds, dg = ctx.alloc(32 * n), ctx.alloc(96 * n), ctx.alloc(32 * n), ctx.alloc(96)
dg.upload(GX.to_bytes(48, "little") + GY.to_bytes(48, "little"))
ctx.gen_scalars_device(dk, n, 1); ctx.batch_mul_device(dg, 1, dk, dp, n); ctx.gen_scalars_device(ds, n, 2)
ref = ctx.msm_device(dp, ds, n)
# same inputs held in torch tensors
tp = torch.frombuffer(bytearray(dp.download()), dtype=torch.uint8).cuda()
ts = torch.frombuffer(bytearray(ds.download()), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
got = ctx.msm_device(tp.data_ptr(), ts.data_ptr(), n)
assert N.cg1_eq(ref, got) == 1
print("torch tensor data_ptr inputs OK; torch still alive:", float((x * 2).sum()))
