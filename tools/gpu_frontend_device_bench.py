#!/usr/bin/env python3
"""The device front-end (k_shuffle_front_end) on a batch of 1024 DISTINCT ell = 124 proofs (tests/golden fixture): latency of one
launch for several transcripts-per-wave settings, throughput with several launches in flight on separate contexts (the kernel
occupies 16 .. 1024 of the chip's 1024 SIMDs at one wave each), against the host front-end on all cores."""
import ctypes, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
from batch_fixture import ShuffleBatch

fx = ShuffleBatch()
ctx = N.Context(0)
v = ShuffleBatchVerifier(fx.crs, ctx)
crs = v.crs
n = 1024
inst, proofs, _ = fx.tiled(n)
L, K = crs.points_per_proof, N.cg1_shuffle_rowin_scalars(crs.handle)
w = v.draw_weights(n)
wire = ctypes.create_string_buffer(n * L * 48)
assert N.cg1_shuffle_gather_points(crs.handle, n, inst, proofs, wire) == 0
aux = ctypes.create_string_buffer(n * 19 * 32)
assert N.cg1_shuffle_gather_aux(crs.handle, n, proofs, w, aux) == 0


class Slot:
    def __init__(self, cx):
        self.ctx = cx
        self.d_wire, self.d_pts, self.d_pst = cx.alloc(n * L * 48), cx.alloc(n * L * 96), cx.alloc(n * L)
        self.d_aux, self.d_rowin, self.d_st = cx.alloc(n * 19 * 32), cx.alloc(n * K * 32), cx.alloc(4 * n)
        self.d_wire.upload(wire.raw); self.d_aux.upload(aux.raw)
        cx.check(N.cg1_batch_decompress_device(cx.handle, self.d_wire.ptr, self.d_pts.ptr, self.d_pst.ptr, n * L, 0))
        self.fe = N.cg1_shuffle_fe_create(cx.handle, crs.ell, crs.lg, crs.affine96, crs.bytes)
        assert self.fe

    def run(self, lanes):
        self.ctx.check(N.cg1_shuffle_fe_enqueue(self.fe, self.ctx.handle, n, self.d_wire.ptr, self.d_pts.ptr, self.d_aux.ptr, self.d_rowin.ptr, self.d_st.ptr, lanes))
        self.ctx.check(N.cg1_stream_sync(self.ctx.handle))


s0 = Slot(ctx)
# reference: the host front-end on all cores
pts = s0.d_pts.download()
decoded = b"".join(pts[(i * L + 4 * crs.ell + 1) * 96: (i * L + 4 * crs.ell + 9) * 96] for i in range(n))
h_pts, h_rowin, h_status = ctypes.create_string_buffer(n * L * 48), ctypes.create_string_buffer(n * K * 32), (ctypes.c_int32 * n)()
for _ in range(2):
    t = time.perf_counter()
    assert N.cg1_shuffle_prepare_inputs(crs.handle, n, inst, proofs, w, decoded, 768, h_pts, h_rowin, h_status, 0) == 0
    host_ms = (time.perf_counter() - t) * 1e3
print(f"host front-end (cg1_shuffle_prepare_inputs, {int(N.cg1_shuffle_default_threads())} threads): {host_ms:.2f} ms per {n} proofs", flush=True)
print(f"block program: {N.cg1_shuffle_fe_nodes(s0.fe)} nodes", flush=True)
for fe_rows, timed in ((1, 0), (1, 1), (0, 0)):
    ctx.set_param("fe_rows", fe_rows)
    ctx.set_param("fe_timed", timed)
    for lanes in ((64, 16, 4, 1) if not timed else (64,)):
        s0.run(lanes)
        t = time.perf_counter(); s0.run(lanes); dt = (time.perf_counter() - t) * 1e3
        passes = N.cg1_shuffle_fe_last_passes(s0.fe, ctx.handle)
        sp = (ctypes.c_uint32 * 7)()
        N.cg1_shuffle_fe_last_split(s0.fe, sp)
        split = " (kclk per pass: pieces+loads %.1f, keccak %.1f, whole %.1f, draw %.1f; steps: gprod %.0f + D/A' %.0f + final %.0f kclk)" % tuple([256e-3 * x / max(1, passes) for x in sp[:4]] + [256e-3 * x for x in sp[4:7]]) if passes and timed else ""
        ok = s0.d_rowin.download() == h_rowin.raw and list((ctypes.c_int32 * n).from_buffer_copy(s0.d_st.download())) == list(h_status)
        print(f"device front-end ({'block program' if fe_rows else 'byte machine'}), {lanes:2d} transcripts per wave ({(n + lanes - 1) // lanes} waves): {dt:.2f} ms per launch of {n} "
              f"proofs; passes of the slowest wave: {passes}{split}; blocks equal the host's: {ok}", flush=True)
ctx.set_param("fe_rows", 1); ctx.set_param("fe_timed", 0)
slots = [s0] + [Slot(N.Context(0)) for _ in range(5)]
for lanes in (64, 16):
    for k in (2, 4, 6):
        reps = 3
        def work(s):
            for _ in range(reps):
                s.run(lanes)
        ths = [threading.Thread(target=work, args=(s,)) for s in slots[:k]]
        t = time.perf_counter()
        for th in ths: th.start()
        for th in ths: th.join()
        dt = (time.perf_counter() - t) * 1e3
        print(f"{k} launches in flight ({lanes} per wave): {dt / (reps * k):.2f} ms per batch of {n} = {n * reps * k / dt * 1e3:.0f} proofs/s of front-end", flush=True)
