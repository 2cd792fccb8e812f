#!/usr/bin/env python3
"""Single-rank RCCL self-test THROUGH THE C ABI (a one-GPU box cannot host two RCCL ranks): cg1_comm_create(0, 1) +
cg1_comm_attach_rccl (dlopen librccl.so, ncclGetUniqueId, ncclCommInitRank on the context's device), then the exact
collectives bench.py / distributed.py issue at N > 1 -- cg1_comm_allreduce_g1 (ncclAllGather of a 144-byte blob on the
context's compute stream), a generic byte all-gather, the control-channel barrier and clock gather -- next to a live MSM on the
same context.  No PyTorch."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N
from curdleproofs_pie_amd.distributed import all_reduce_g1, init_comm, max_over_ranks

assert "torch" not in sys.modules
ctx = N.Context(0)
comm = init_comm(0, 1)
t0 = time.perf_counter()
comm.attach_rccl(ctx)
t_init = time.perf_counter() - t0
assert comm.transport == "rccl" and comm.world_seen == 1          # ncclCommCount
g = ctypes.create_string_buffer(N.POINT_BYTES); N.cg1_generator(g)
assert all_reduce_g1(g.raw, comm) == g.raw                          # world = 1 short-cut in Python ...
assert N.cg1_eq(comm.allreduce_g1(g.raw), g.raw)                    # ... and the real ncclAllGather path
payload = bytes(range(200)) * 3
assert comm.allgather(payload) == [payload]
comm.barrier()
assert max_over_ranks(1.25, comm) == [1.25]
# an MSM on the context's stream between collectives
d = ctx.alloc(96); aff = ctypes.create_string_buffer(96); N.cg1_to_affine96(aff, g.raw); d.upload(aff.raw)
s = ctx.alloc(32); s.upload((5).to_bytes(32, "little"))
blob = ctx.msm_device(d, s, 1)
five = ctypes.create_string_buffer(N.POINT_BYTES); N.cg1_mul(five, g.raw, (5).to_bytes(32, "little"))
assert N.cg1_eq(blob, five.raw) and N.cg1_eq(comm.allreduce_g1(blob), five.raw)
reps = 200
t0 = time.perf_counter()
for _ in range(reps):
    comm.allreduce_g1(blob)
us = (time.perf_counter() - t0) / reps * 1e6
assert "torch" not in sys.modules
comm.close()
print("rccl selftest ok (C ABI, no torch): ncclCommInitRank %.2f s, cg1_comm_allreduce_g1 %.1f us per call at world 1" % (t_init, us))
