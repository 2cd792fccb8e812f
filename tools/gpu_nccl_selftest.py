#!/usr/bin/env python3
"""Single-rank RCCL self-test of the exact collectives bench.py / distributed.py issue at N > 1 (a one-GPU box cannot
host two NCCL ranks): uint8 all_gather_into_tensor of a 144-byte blob, int32 all_gather, barrier, float64 MAX all-reduce,
next to a live MSM context on the same device."""
import os, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
import torch.distributed as dist
from curdleproofs_pie_amd import _native as N

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
ctx = N.Context(0)
g = ctypes.create_string_buffer(N.POINT_BYTES); N.cg1_generator(g)
mine = torch.empty(N.POINT_BYTES, dtype=torch.uint8, device="cuda")
gathered = torch.empty(N.POINT_BYTES, dtype=torch.uint8, device="cuda")
mine.copy_(torch.frombuffer(bytearray(g.raw), dtype=torch.uint8))
dist.all_gather_into_tensor(gathered, mine)
assert gathered.cpu().numpy().tobytes() == g.raw
local = torch.tensor([0, 6, 1], dtype=torch.int32, device="cuda")
out = [torch.empty_like(local)]
dist.all_gather(out, local)
assert out[0].cpu().tolist() == [0, 6, 1]
dist.barrier(); torch.cuda.synchronize(); ctx.sync()
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
# an MSM on the context's own stream while torch holds the device
d = ctx.alloc(96); aff = ctypes.create_string_buffer(96); N.cg1_to_affine96(aff, g.raw); d.upload(aff.raw)
s = ctx.alloc(32); s.upload((5).to_bytes(32, "little"))
blob = ctx.msm_device(d, s, 1)
five = ctypes.create_string_buffer(N.POINT_BYTES); N.cg1_mul(five, g.raw, (5).to_bytes(32, "little"))
assert N.cg1_eq(blob, five.raw)
dist.barrier()
dist.destroy_process_group()
print("rccl selftest ok")
