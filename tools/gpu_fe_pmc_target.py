#!/usr/bin/env python3
"""A short target for `rocprofv3 --pmc ...` / `--kernel-trace --stats`: three launches of the device front-end as a block program
(k_fill_rows + k_shuffle_front_end_rows), one of the byte-level machine (k_shuffle_front_end), on the 1024 distinct fixture proofs."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
from batch_fixture import ShuffleBatch

fx = ShuffleBatch()
ctx = N.Context(0)
v = ShuffleBatchVerifier(fx.crs, ctx, device_front_end=False)
crs = v.crs
n = 1024
inst, proofs, _ = fx.tiled(n)
L, K = crs.points_per_proof, N.cg1_shuffle_rowin_scalars(crs.handle)
w = v.draw_weights(n)
wire = ctypes.create_string_buffer(n * L * 48)
assert N.cg1_shuffle_gather_points(crs.handle, n, inst, proofs, wire) == 0
aux = ctypes.create_string_buffer(n * 19 * 32)
assert N.cg1_shuffle_gather_aux(crs.handle, n, proofs, w, aux) == 0
d_wire, d_pts, d_pst = ctx.alloc(n * L * 48), ctx.alloc(n * L * 96), ctx.alloc(n * L)
d_aux, d_rowin, d_st = ctx.alloc(n * 19 * 32), ctx.alloc(n * K * 32), ctx.alloc(4 * n)
d_wire.upload(wire.raw); d_aux.upload(aux.raw)
ctx.check(N.cg1_batch_decompress_device(ctx.handle, d_wire.ptr, d_pts.ptr, d_pst.ptr, n * L, 0))
fe = N.cg1_shuffle_fe_create(ctx.handle, crs.ell, crs.lg, crs.affine96, crs.bytes)
for fe_rows in (1, 1, 1, 0):
    ctx.set_param("fe_rows", fe_rows)
    ctx.check(N.cg1_shuffle_fe_enqueue(fe, ctx.handle, n, d_wire.ptr, d_pts.ptr, d_aux.ptr, d_rowin.ptr, d_st.ptr, 0))
    ctx.check(N.cg1_stream_sync(ctx.handle))
    print("fe_rows", fe_rows, "passes", N.cg1_shuffle_fe_last_passes(fe, ctx.handle), flush=True)
