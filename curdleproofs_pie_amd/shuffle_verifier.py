"""Batch verification of Whisk shuffle proofs: many `IsValidWhiskShuffleProof` calls as ONE GPU MSM.

Stands behind `IsValidWhiskShuffleProof` / `AssertIsValidWhiskShuffleProof`
(curdleproofs/curdleproofs/whisk_interface.py:72-109) and `CurdleProofsProof.verify` (curdleproofs.py:160-246) for
BATCHES of proofs over one CRS -- BASELINE configs 3 and 5 (1 024 / 16 384 ell=128 verifications).  Same inputs
(tracker encodings, proof bytes, CRS), same verdicts; the work is re-cut for the machine:

  host, native, one thread per core (csrc/shuffle_verify.cpp):  wire parsing, the Fiat-Shamir transcript, all Fr
      arithmetic; every verifier equation of a proof becomes a row of scalars over the proof's own wire points and
      over the CRS points (left-hand sides included, each check weighted by a fresh random rho);
  GPU:  batched decompression of every wire point (k_batch_decompress), then
      mode "merged":       ONE regime-A Pippenger MSM over  CRS (scalars summed across proofs) + all own points;
                           identity  <=>  every prepared proof verifies (soundness error ~2^-250 per batch);
                           on failure falls back to "independent" to name the culprits;
      mode "independent":  one regime-B MSM per proof (own points) + one per proof over the CRS points.

There is no CPU path for the group arithmetic: without a GPU `Context` creation raises NativeError.
The reference draws its rho from Python's global `random` (util.py:21-24, msm_accumulator.py:43); a batch
verifier must not be predictable, so weights come from `secrets` unless the caller passes `rng` (tests do).
"""
from __future__ import annotations

import ctypes
import os
import secrets
from typing import List, Optional, Sequence, Tuple

from . import _native as N

FR_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # util.py:7
N_BLINDERS = 4                                                                   # curdleproofs.py:26
N_WEIGHTS = 12
N_EXACT_POINTS = 10            # T_1 T_2 U_1 U_2 R S cm_A cm_B: the proof's points in the same-scalar equalities (same_scalar.py:82-108)

REJECT_NAMES = {0: "prepared", 1: "bad scalar encoding", 2: "bad point encoding", 3: "vec_T[0] is infinity", 4: "bad weight",
                5: "bad length", 6: "verification equation failed"}
REJECT_LENGTH, REJECT_EQUATION = 5, 6
_CLEAR_TOP2 = bytes(b & 0x3F for b in range(256))


def _addr(b) -> int:
    """Address of the first byte of a bytes object / ctypes buffer (to pass sub-ranges of a batch to native code)."""
    if isinstance(b, bytes):
        return ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p).value
    return ctypes.addressof(b)


class _Segments:
    """Several callers' batches travelling as ONE internal batch without being copied together: item i of the internal batch is item
    i - first of the segment that holds it.  `runs(lo, hi)` cuts [lo, hi) at the segment boundaries."""

    def __init__(self, parts, item_bytes: int):
        self.parts = list(parts)                              # [(bytes-like, n_items)]
        self.item_bytes = item_bytes
        self.first = [0]
        for _, n in self.parts:
            self.first.append(self.first[-1] + n)

    def __len__(self) -> int:
        return self.first[-1] * self.item_bytes

    def runs(self, lo: int, hi: int):
        """(address of item `a`, a, count) for every maximal run [a, a + count) of [lo, hi) inside one segment"""
        for k, (buf, n) in enumerate(self.parts):
            f = self.first[k]
            a, b = max(lo, f), min(hi, f + n)
            if a < b:
                yield _addr(buf) + (a - f) * self.item_bytes, a, b - a

    def item_addr(self, i: int) -> int:
        for addr, a, cnt in self.runs(i, i + 1):
            return addr
        raise IndexError(i)

    def joined(self) -> bytes:
        return b"".join(bytes(buf) for buf, _ in self.parts)


def _runs(buf, item_bytes: int, lo: int, hi: int):
    """The same for a plain contiguous buffer: one run."""
    if isinstance(buf, _Segments):
        yield from buf.runs(lo, hi)
    else:
        yield _addr(buf) + lo * item_bytes, lo, hi - lo


def _tracker_bytes(trackers) -> Tuple[bytes, bytes]:
    """Sequence of WhiskTracker-likes (r_G, k_r_G attributes; whisk_interface.py:24-30) or (r_G, k_r_G) pairs.
    Every encoding must be exactly 48 bytes (the reference decodes them one by one, whisk_interface.py:96-100, and raises on
    any other length): adjacent encodings of 47 and 49 bytes must not be re-split at 48-byte boundaries."""
    trackers = list(trackers)
    if trackers and hasattr(trackers[0], "r_G"):
        rs, ks = [t.r_G for t in trackers], [t.k_r_G for t in trackers]
    else:
        rs, ks = [t[0] for t in trackers], [t[1] for t in trackers]
    if not all(len(x) == 48 for x in rs) or not all(len(x) == 48 for x in ks):
        raise ValueError("tracker encoding is not 48 bytes")
    try:
        return b"".join(rs), b"".join(ks)                 # bytes-likes (BLSPubkey is a bytes subclass)
    except TypeError:
        return b"".join(bytes(r) for r in rs), b"".join(bytes(k) for k in ks)


class ShuffleCrs:
    """The CRS in wire form (CurdleproofsCrs.to_bytes, crs.py:92-101) plus what the native front-end derives from it."""

    def __init__(self, crs, ell: Optional[int] = None):
        data = bytes(crs.to_bytes()) if hasattr(crs, "to_bytes") else bytes(crs)
        if ell is None:
            ell = len(data) // 48 - N_BLINDERS - 5
        if len(data) != 48 * (ell + N_BLINDERS + 5):
            raise ValueError("CRS bytes do not match ell")
        self.ell = ell
        self.bytes = data
        self.handle = N.cg1_shuffle_crs_create(data, ell, N_BLINDERS)
        if not self.handle:
            raise ValueError("invalid CRS (sizes: ell + 4 must be a power of two; every point must decode)")
        self.proof_bytes = N.cg1_shuffle_proof_bytes(self.handle)
        self.points_per_proof = N.cg1_shuffle_points_per_proof(self.handle)
        self.ncrs = N.cg1_shuffle_crs_points(self.handle)
        self.challenges_per_proof = N.cg1_shuffle_challenges_per_proof(self.handle)
        self.lg = (ell + N_BLINDERS).bit_length() - 1
        blobs = ctypes.create_string_buffer(N.POINT_BYTES * self.ncrs)
        bad = ctypes.c_size_t(0)
        if N.cg1_batch_decompress(blobs, data, self.ncrs, 0, ctypes.byref(bad)):
            raise ValueError("invalid CRS point")
        aff = ctypes.create_string_buffer(96 * self.ncrs)
        N.cg1_batch_to_affine96(aff, blobs.raw, self.ncrs)
        self.affine96 = aff.raw

    def __del__(self):
        try:
            if self.handle:
                N.cg1_shuffle_crs_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Prepared:
    """Output of the native front-end for n proofs (host buffers)."""

    def __init__(self, crs: ShuffleCrs, n: int, want_challenges: bool, staging=None):
        L, C = crs.points_per_proof, crs.ncrs
        self.n = n
        if staging is not None:                    # page-locked buffers of the GPU flow (scalars32 has room for the CRS row)
            self.points48, self.scalars32 = staging["wire"].buf, staging["sc"].buf
        else:
            self.points48 = ctypes.create_string_buffer(max(1, n * L * 48))
            self.scalars32 = ctypes.create_string_buffer(max(1, n * L * 32))
        if staging is not None and staging.get("crs_scalars") is not None and len(staging["crs_scalars"]) >= n * C * 32:
            self.crs_scalars32, self.status = staging["crs_scalars"], staging["status"]      # every byte is rewritten by prepare
        else:
            self.crs_scalars32 = ctypes.create_string_buffer(max(1, n * C * 32))
            self.status = (ctypes.c_int32 * max(1, n))()
            if staging is not None:
                staging["crs_scalars"], staging["status"] = self.crs_scalars32, self.status
        self.challenges = ctypes.create_string_buffer(n * crs.challenges_per_proof * 32) if want_challenges and n else None


class ShuffleBatchVerifier:
    SPLIT = 2048                            # verify_packed cuts calls of more than 2 * SPLIT proofs into pieces of this many
    FE_FIRST_MAX = 8                        # host front-end: batches up to this size run it BESIDE the GPU's decoding (it decodes its own four points)

    def __init__(self, crs, ctx: Optional["N.Context"] = None, threads: int = 0, chunk: int = 256, device_rows: bool = True,
                 blocking_sync: Optional[bool] = None, device_front_end: Optional[bool] = None, fe_lanes: Optional[int] = None, fe_cus: int = 0, fe_prio: int = 0,
                 pipelines: Optional[int] = None, max_pinned_bytes: Optional[int] = None, coalesce: Optional[int] = None):
        # max_pinned_bytes: budget for the page-locked staging of the batch slots of ONE pipeline (None = no limit).  A stream keeps
        # 3 + fe_lanes + 1 slots rotating so that packing, decoding, front-end launches and the MSM of different batches overlap; under a
        # budget the rotation shrinks towards the minimum of 3 (fewer batches in flight: throughput cost in profiles/r05_verify_footprint.txt).
        # The other knob is `pipelines` (each is a complete set of contexts, threads and slots).  footprint() reports what is held.
        self.max_pinned_bytes = max_pinned_bytes
        self._coalesce_arg = coalesce
        self.crs = crs if isinstance(crs, ShuffleCrs) else ShuffleCrs(crs)
        self._ctx = ctx
        self.threads = threads
        # blocking_sync: the two GPU-lane threads wait for the device asleep (hipEventBlockingSync) instead of spinning.  Measured
        # neutral on a 16-thread box (profiles/r03_verify_thread_sweep.txt: the front-end is bound by its 8 physical cores either
        # way), so the default stays the lower-latency spinning wait; CURDLE_G1_BLOCKING_SYNC=1 turns it on.
        env = os.environ.get("CURDLE_G1_BLOCKING_SYNC")
        self.blocking_sync = (env == "1") if env is not None else bool(blocking_sync)
        self.chunk = chunk                  # sub-batch of the decompress / front-end pipeline
        # device_front_end: the transcript, D / A' and the challenge algebra run on the GPU too (csrc/kernels_frontend.h, one proof per
        # lane; byte-identical row-input blocks: tests/test_shuffle_frontend_gpu.py).  One launch takes ~11 ms whatever its size (a
        # transcript is ~800 dependent Keccak permutations: 24 K clocks each for the one wave a SIMD runs) but occupies only n / 64 of
        # the chip's 1024 SIMDs, so `fe_lanes` launches of consecutive batches run side by side, each on its own context, and the
        # stream keeps fe_lanes + 3 batches in flight.  The host then only packs bytes: proofs/s no longer depends on its core count.
        # Measured on MI355X (profiles/r03_verify_fe_ab.txt, profiles/r03_v7_bench.json; batches of 1024, GPU_MAX_HW_QUEUES = 24):
        # 158-167 K proofs/s on 2 or 4 host threads with three pipelines (below), whatever the box; host front-end: 13 K / 25 K / 49 K /
        # 91 K proofs/s on 1 / 2 / 4 / 8 threads and 124-168 K on 16, depending on the box's cores -- so None (the default) turns the
        # device front-end on when fewer than twenty-four host threads are available to this verifier and leaves the host front-end on
        # otherwise (a single isolated batch comes back sooner from the host: 19 against ~45 ms).  CURDLE_G1_DEVICE_FRONT_END=0/1 overrides.
        env = os.environ.get("CURDLE_G1_DEVICE_FRONT_END")
        if env in ("0", "1"):
            device_front_end = env == "1"
        self._fe_by_default = device_front_end is None      # nobody asked for one of the two: an ISOLATED batch may take the other (verify_packed)
        self._host_twin = None
        if device_front_end is None:
            # (round 5: also with the runtime's default of 4 hardware queues -- a stream then travels in coalesced internal batches of up to
            # 4 096 proofs, _verify_stream_coalesced: 1.55e5 proofs/s whatever the host's cores, against 7.6e4 for 1 024-proof batches alone;
            # profiles/r05_verify_queues_ab.txt, r05_bench_verify_4_queues.json)
            device_front_end = (threads or int(N.cg1_shuffle_default_threads())) < 24
        self.device_front_end = bool(device_front_end)
        # pipelines (device front-end only): that many complete pipelines -- decoding lane, front-end launches, MSM lane, each on contexts
        # of its own -- take the batches of a stream in turn.  One pipeline leaves the GPU idle between its dependent kernels (the
        # reduce chains of an MSM, the waits of a decoding lane): 7.0-7.8 ms per batch of 1024; two 6.4-6.6; three 6.16-6.33 (1.62-1.66e5
        # proofs/s); four are worse again (6.9-8.0: the process runs out of hardware queues) -- profiles/r03_verify_fe_ab.txt.  With the
        # host front-end a second pipeline only fights for the cores (11 ms), so it stays 1.
        # The default follows the hardware queues the process has (N.hw_queues(): GPU_MAX_HW_QUEUES, 4 unless the application raised it
        # -- N.tune_runtime()): a pipeline keeps 4 streams busy, so 3 pipelines from 20 queues, 2 from 12, else ONE pipeline with two
        # front-end launches (streams beyond the queue count share queues and wait for each other's kernels).
        if pipelines is None:
            q = N.hw_queues()
            pipelines = (3 if q >= 20 else 2 if q >= 12 else 1) if self.device_front_end else 1
        self.pipelines = max(1, int(pipelines)) if self.device_front_end else 1
        if fe_lanes is None:
            fe_lanes = 2 if (self.pipelines > 1 or N.hw_queues() < 8) else 3
        self._kids = None
        self.fe_lanes = max(1, int(fe_lanes)) if self.device_front_end else 0
        # coalesce: a STREAM of batches is verified in internal batches of up to this many proofs (_verify_stream_coalesced; 0 = never): a
        # front-end launch takes ~12 ms whatever it carries, so larger internal batches amortise it.  Measured with 1 024-proof batches
        # (profiles/r05_verify_queues_ab.txt): 4 hardware queues (the runtime's default; one pipeline) 8.2e4 -> 1.55e5 proofs/s at 4 096;
        # 24 queues (three pipelines) 1.72e5 -> 1.86e5 at 2 048 (4 096 starves the pipelines: 1.51e5).
        if self._coalesce_arg is None:
            self.coalesce = (4096 if N.hw_queues() < 12 else 2048) if self.device_front_end else 0
        else:
            self.coalesce = max(0, int(self._coalesce_arg))
        # fe_cus > 0 (A/B switch, off): the front-end launches get that many compute units of their own (the last ones) and the
        # throughput kernels are confined to the others (hipExtStreamCreateWithCUMask).  Measured a loss: the masked decompression
        # stream took 30 ms instead of 6 per batch (profiles/r03_verify_fe_ab.txt)
        self.fe_cus = int(fe_cus) if self.device_front_end else 0
        self.fe_prio = int(fe_prio)
        self._cu_total = 256
        if self.fe_cus and self._ctx is not None:
            self._user_ctx = self._ctx                 # the caller's context stays untouched: this mode drives contexts of its own
            self._ctx = N.Context(self._user_ctx.device, cu_mask=range(0, self._cu_total - self.fe_cus))
            self._own_ctx = True
        self._gpu_threads = [None, None] + [None] * self.fe_lanes
        self._gpu_jobs = [None, None] + [None] * self.fe_lanes
        self._fe = [None] * self.fe_lanes              # (context, cg1_shuffle_fe handle) per front-end lane
        self._fe_next = 0
        self._ctx_msm = None
        self.prefetch_big = True            # two large decompress launches for batches decoded a batch ahead (A/B switch)
        # True: the host front-end emits only the challenges (+ a few derived scalars) per proof and the GPU expands them into
        # the scalar rows (k_shuffle_rows: SURVEY 8(f) row 3) -- 5.7 KB instead of 23 KB per proof over PCIe, and a quarter
        # of the front-end's arithmetic off the host.  False: rows on the host (A/B switch; what `prepare()` always does).
        self.device_rows = device_rows
        self._rowin_scalars = N.cg1_shuffle_rowin_scalars(self.crs.handle)
        self._nslots = 3 + self.fe_lanes + (1 if self.fe_lanes else 0)
        self._slots = [None] * self._nslots
        self._next_slot = 0
        self.last_stats = {}
        self.last_status = []
        ell, lg = self.crs.ell, self.crs.lg
        offs = [4 * ell + 2 + j for j in range(6)] + [4 * ell + 12 + 4 * lg + j for j in range(4)]
        self._exact_offsets = (ctypes.c_uint32 * N_EXACT_POINTS)(*offs)

    # ---------------------------------------------------------------- lifetime
    def close(self) -> None:
        """Stop the two GPU threads (after what is queued has drained) and release the buffer slots and the MSM context.
        The verifier must not be used afterwards.  Idempotent; also run by __del__ and on cache eviction."""
        self._release_lanes()
        if getattr(self, "_host_twin", None) is not None:
            self._host_twin.close()
            self._host_twin = None
        for kid in (getattr(self, "_kids", None) or []):
            kid.close()
        self._kids = None
        if getattr(self, "_ctx_blocking", False) and self._ctx is not None and self._ctx.handle:
            self._ctx.set_param("blocking_sync", 0)
            self._ctx_blocking = False
        if getattr(self, "_own_ctx", False) and self._ctx is not None:
            self._ctx.close()
            self._ctx = None
            self._own_ctx = False

    def _release_lanes(self) -> None:
        """Give back what this verifier's OWN lanes hold -- GPU threads, buffer slots, the MSM context, the front-end contexts -- and
        with them their hardware queues (a process has 24: a verifier whose pipelines run on child verifiers must not keep lanes of its
        own alive beside them).  They are rebuilt on demand."""
        for lane in range(len(self._gpu_threads)):
            t, q = self._gpu_threads[lane], self._gpu_jobs[lane]
            if t is not None:
                q.put(None)
                t.join()
            self._gpu_threads[lane] = self._gpu_jobs[lane] = None
        for i, b in enumerate(self._slots):
            if b is not None:
                for k in ("wire", "pts", "pstat", "sgflags", "sc", "rowin", "hstat", "csrows", "dstat"):
                    b[k].free()
                for h in b["host"].values():
                    if isinstance(h, N.PinnedBuffer):          # (the staging dict also caches plain ctypes buffers)
                        h.free()
                self._slots[i] = None
        if self._ctx_msm is not None:
            self._ctx_msm.close()
            self._ctx_msm = None
        for k, pair in enumerate(getattr(self, "_fe", [])):
            if pair is not None:
                cx, fe, aux_h, aux_d = pair
                N.cg1_shuffle_fe_destroy(fe)
                aux_h.free(); aux_d.free()
                cx.close()
                self._fe[k] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- host half
    def pack(self, items) -> Tuple[bytes, bytes, List[int]]:
        """items: (pre_trackers, post_trackers, proof_bytes) triples -> fixed-stride instance / proof buffers.
        Items of the wrong shape get a REJECT_LENGTH verdict (the reference raises on them: the list
        comprehensions / BufReader of whisk_interface.py:96-106 run out of data) and a zero-filled slot."""
        ell, pb = self.crs.ell, self.crs.proof_bytes
        inst, proofs, pre_status = [], [], []
        for pre, post, proof in items:
            proof = bytes(proof)
            try:
                pr, pk = _tracker_bytes(pre)
                qr, qk = _tracker_bytes(post)
            except Exception:
                pr = pk = qr = qk = b""
            ok = (len(pr) == len(pk) == len(qr) == len(qk) == 48 * ell and len(proof) >= pb
                  and len(pre) == len(post) == ell)
            if ok:
                inst.append(pr + pk + qr + qk)        # vec_R | vec_S | vec_T | vec_U
                proofs.append(proof[:pb])             # trailing bytes are never read by BufReader (util.py:138-153)
                pre_status.append(0)
            else:
                inst.append(bytes(4 * ell * 48))
                proofs.append(bytes(pb))
                pre_status.append(REJECT_LENGTH)
        return b"".join(inst), b"".join(proofs), pre_status

    def draw_weights(self, n: int, rng=None) -> bytes:
        """12 weights per proof.  Default: 254 uniformly random bits each from the OS CSPRNG in one call (a cheating
        proof survives a batch with probability ~2^-254); with `rng` (tests): rng.randint(1, r-1) like util.py:21-24."""
        if rng is None:
            raw = bytearray(secrets.token_bytes(32 * N_WEIGHTS * n))
            raw[31::32] = raw[31::32].translate(_CLEAR_TOP2)            # < 2^254 < r: canonical without rejection
            return bytes(raw)
        return b"".join(rng.randint(1, FR_MODULUS - 1).to_bytes(32, "little") for _ in range(N_WEIGHTS * n))

    def prepare(self, instances: bytes, proofs: bytes, n: int, weights: Optional[bytes] = None, rng=None,
                want_challenges: bool = False, decoded=None, staging=None) -> Prepared:
        """Native front-end for n packed proofs.  `decoded`: per proof the 8 own points 4*ell+1 .. 4*ell+8 as the GPU
        decompressed them (768 B each); None = the host decodes the four it needs itself."""
        crs = self.crs
        assert len(instances) == n * 4 * crs.ell * 48 and len(proofs) == n * crs.proof_bytes
        if weights is None:
            weights = self.draw_weights(n, rng)
        assert len(weights) == n * N_WEIGHTS * 32
        assert decoded is None or len(decoded) >= n * 768
        out = Prepared(crs, n, want_challenges, staging)
        rc = N.cg1_shuffle_prepare(crs.handle, n, instances, proofs, weights, decoded, 768, out.points48, out.scalars32,
                                   out.crs_scalars32, out.status, out.challenges, self.threads)
        if rc:
            raise N.NativeError(f"cg1_shuffle_prepare failed ({rc})")
        return out

    # ---------------------------------------------------------------- GPU half
    @property
    def ctx(self) -> "N.Context":
        if self._ctx is None:
            if self.fe_cus:
                self._ctx = N.Context(N.default_context().device, cu_mask=range(0, self._cu_total - self.fe_cus))
                self._own_ctx = True
            else:
                self._ctx = N.default_context()
        if self.blocking_sync and not getattr(self, "_ctx_blocking", False):
            self._ctx.set_param("blocking_sync", 1)        # (a shared context: close() sets it back)
            self._ctx_blocking = True
        return self._ctx

    @property
    def ctx_msm(self) -> "N.Context":
        """A second context on the same GPU for the MSM stage: its kernels, its host Horner tail and its dependent reduce
        kernels then overlap the decompression of the next batch (which runs on `ctx` from another thread)."""
        if self._ctx_msm is None:
            self._ctx_msm = N.Context(self.ctx.device, cu_mask=range(0, self._cu_total - self.fe_cus) if self.fe_cus else None)
            self._ctx_msm.set_param("blocking_sync", 1 if self.blocking_sync else 0)
        return self._ctx_msm

    def _gpu_submit(self, fn, lane: int = 0) -> None:
        """GPU work runs on two persistent threads (a thread's first HIP call is expensive; a context takes one call at
        a time): lane 0 drives `ctx` (staging + decompression, in submission order), lane 1 drives `ctx_msm` (MSMs)."""
        import queue
        import threading

        if self._gpu_threads[lane] is None:
            jobs = self._gpu_jobs[lane] = queue.Queue()

            def loop(jobs=jobs):
                while True:
                    job = jobs.get()
                    if job is None:
                        return
                    job()

            self._gpu_threads[lane] = threading.Thread(target=loop, daemon=True)
            self._gpu_threads[lane].start()
        self._gpu_jobs[lane].put(fn)

    def _alloc_slot(self, n: int) -> dict:
        crs, ctx = self.crs, self.ctx
        L, C = crs.points_per_proof, crs.ncrs
        return {
            "cap": n, "busy": None,
            "wire": ctx.alloc(n * L * 48),
            "pts": ctx.alloc((n * L + C) * 96),              # own points of all proofs, then the CRS points
            "pstat": ctx.alloc(n * L),
            "sgflags": ctx.alloc(n * N_EXACT_POINTS),        # 1 = outside G1, for the points of the exactly-asserted equalities
            "sc": ctx.alloc((n * L + C) * 32),
            "rowin": ctx.alloc(n * self._rowin_scalars * 32),   # device row builder: input blocks, host codes, CRS rows, final codes
            "hstat": ctx.alloc(4 * n), "csrows": ctx.alloc(n * C * 32), "dstat": ctx.alloc(4 * n),
            "host": {                                        # page-locked staging
                "wire": N.PinnedBuffer(ctx, n * L * 48),
                "sc": N.PinnedBuffer(ctx, (n * L + C) * 32),
                "rowin": N.PinnedBuffer(ctx, n * self._rowin_scalars * 32),
                "hstat": N.PinnedBuffer(ctx, 4 * n),
                "pstat": N.PinnedBuffer(ctx, n * L),
                "decoded": N.PinnedBuffer(ctx, n * 768),
                "fe_wire": N.PinnedBuffer(ctx, min(n, self.FE_FIRST_MAX) * L * 48),   # the front-end's own copy of a tiny batch's points
            },
        }

    def _slot_pinned_bytes(self, n: int) -> int:
        L, C = self.crs.points_per_proof, self.crs.ncrs
        return n * L * 48 + (n * L + C) * 32 + n * self._rowin_scalars * 32 + 4 * n + n * L + n * 768 + min(n, self.FE_FIRST_MAX) * L * 48

    def footprint(self) -> dict:
        """Bytes this verifier holds for its batch slots and front-end lanes: {"pinned_bytes", "device_bytes", "slots", "pipelines": [...]}
        (page-locked host staging / hipMalloc'd buffers; the contexts' own scratch -- MSM buckets, sort buffers -- is not counted here)."""
        pinned = device = slots = 0
        for b in self._slots:
            if b is None:
                continue
            slots += 1
            for v in b.values():
                if isinstance(v, N.DeviceBuffer):
                    device += v.nbytes
            for v in b["host"].values():
                if isinstance(v, N.PinnedBuffer):
                    pinned += v.nbytes
        for pair in self._fe:
            if pair is not None:
                pinned += pair[2].nbytes
                device += pair[3].nbytes
        kids = [k.footprint() for k in (self._kids or [])]
        return {"pinned_bytes": pinned + sum(k["pinned_bytes"] for k in kids), "device_bytes": device + sum(k["device_bytes"] for k in kids),
                "slots": slots, "pipelines": kids}

    def _slot(self, n: int) -> dict:
        """Buffers of one batch in flight (3 slots rotate: MSM of batch k-1, front-end of k, decompression of k+1).
        All three are allocated together the first time a batch size is seen (page-locking tens of MB takes
        milliseconds: not something to meet again in the middle of a stream)."""
        if self.max_pinned_bytes is not None and all(x is None for x in self._slots):
            fit = max(3, int(self.max_pinned_bytes) // max(1, self._slot_pinned_bytes(n)))
            if fit < self._nslots:                            # (decided once, before the first slot exists)
                self._nslots = fit
                self._slots = [None] * fit
                self._next_slot = 0
        idx = self._next_slot
        self._next_slot = (idx + 1) % self._nslots
        b = self._slots[idx]
        if b is not None and b["busy"] is not None:
            b["busy"].wait()                                  # its previous batch must have left the GPU
        if b is None or b["cap"] < n:
            for j in range(self._nslots):
                o = self._slots[j]
                if o is None or (o["cap"] < n and (o["busy"] is None or o["busy"].is_set())):
                    self._slots[j] = self._alloc_slot(n)
            b = self._slots[idx]
            if b["cap"] < n:                                  # was still busy above: wait, then replace
                b["busy"].wait()
                b = self._slots[idx] = self._alloc_slot(n)
        return b

    def _stage_in(self, b: dict, instances, proofs, lo: int, hi: int) -> None:
        """GPU thread: gather the own points of proofs [lo, hi) into page-locked memory and queue their H2D copy on the
        context's copy stream (it overlaps the decompression kernel of the previous sub-batch)."""
        crs, ctx = self.crs, self.ctx
        L = crs.points_per_proof
        wire = b["host"]["wire"].ptr + lo * L * 48
        for (ia, a, cnt), (pa, _, _) in zip(_runs(instances, 4 * crs.ell * 48, lo, hi), _runs(proofs, crs.proof_bytes, lo, hi)):
            ctx.check(N.cg1_shuffle_gather_points(crs.handle, cnt, ia, pa, b["host"]["wire"].ptr + a * L * 48))
        ctx.check(N.cg1_h2d_async(ctx.handle, b["wire"].ptr + lo * L * 48, wire, (hi - lo) * L * 48))

    def _decompress_launch(self, b: dict, lo: int, hi: int) -> None:
        """GPU thread: once the sub-batch's copy has landed, launch its decompression (points stay on the device for the MSM)."""
        ctx, L = self.ctx, self.crs.points_per_proof
        ctx.check(N.cg1_copy_fence(ctx.handle))
        ctx.check(N.cg1_batch_decompress_enqueue(ctx.handle, b["wire"].ptr + lo * L * 48, b["pts"].ptr + lo * L * 96,
                                                 b["pstat"].ptr + lo * L, (hi - lo) * L, 0))

    def _decompress_collect(self, b: dict, lo: int, hi: int) -> None:
        """GPU thread: wait for the kernel, bring back the per-point verdicts and the 8-point window the front-end wants."""
        crs, ctx = self.crs, self.ctx
        L, h, m = crs.points_per_proof, b["host"], hi - lo
        ctx.check(N.cg1_stream_sync(ctx.handle))                  # this context's kernel only: the MSM context keeps running
        ctx.check(N.cg1_d2h(ctx.handle, h["pstat"].ptr + lo * L, b["pstat"].ptr + lo * L, m * L))
        ctx.check(N.cg1_d2h_2d(ctx.handle, h["decoded"].ptr + lo * 768, 768, b["pts"].ptr + (lo * L + 4 * crs.ell + 1) * 96, L * 96, 768, m))

    def _begin(self, batch, mode: str, rng, prefetched: bool = False) -> dict:
        """Stage 1 of a batch (asynchronous): claim a slot and queue the GPU decompression of its sub-batches.
        prefetched: the batch is decoded a whole batch ahead of its front-end, so small sub-batches buy nothing and
        two large launches fill the GPU better (2 x 2.3 ms instead of 4 x 1.4 ms per 1024 proofs)."""
        import queue
        import threading
        import time

        instances, proofs, n = batch[0], batch[1], batch[2]
        crs = self.crs
        L, C = crs.points_per_proof, crs.ncrs
        assert n >= 1 and len(instances) == n * 4 * crs.ell * 48 and len(proofs) == n * crs.proof_bytes
        weights = batch[4] if len(batch) > 4 else None       # None: drawn on the GPU thread while the first kernel runs
        assert weights is None or len(weights) == n * N_WEIGHTS * 32
        b = self._slot(n)
        tk = {"slot": b, "n": n, "instances": instances, "proofs": proofs, "weights": weights, "mode": mode,
              "pre_status": batch[3] if len(batch) > 3 else None, "chunks": queue.Queue(), "done": threading.Event(),
              "error": None, "t0": time.perf_counter(), "flags_done": threading.Event(), "sgflags": None}
        step = max(self.chunk, (n + 1) // 2) if (prefetched and self.prefetch_big) else self.chunk
        tk["bounds"] = [(lo, min(lo + step, n)) for lo in range(0, n, step)]
        b["busy"] = tk["done"]
        # a handful of proofs (IsValidWhiskShuffleProof is a batch of one): the front-end does not wait for the GPU's decoding -- it decodes
        # the four points it needs itself (4 x 14 us per proof) and runs BESIDE the decompression launch instead of behind it
        tk["fe_first"] = n <= self.FE_FIRST_MAX and not self.device_front_end
        if tk["fe_first"] and tk["weights"] is None:
            tk["weights"] = self.draw_weights(n, rng)

        def gpu_stage(tk=tk, b=b):
            try:
                t_dec = time.perf_counter()
                if b.get("crs_at") != n:                       # CRS points sit right behind the batch's own points
                    b["pts"].upload(crs.affine96, n * L * 96)
                    b["crs_at"] = n
                bounds = tk["bounds"]
                self._stage_in(b, instances, proofs, *bounds[0])
                for i, (lo, hi) in enumerate(bounds):              # copy of sub-batch i+1 overlaps the kernel of sub-batch i
                    self._decompress_launch(b, lo, hi)
                    if tk["weights"] is None:
                        tk["weights"] = self.draw_weights(n, rng)
                    if i + 1 < len(bounds):
                        self._stage_in(b, instances, proofs, *bounds[i + 1])
                    if i + 1 == len(bounds):
                        # beside whatever runs next (side stream, one launch per batch): are the proof's points in the
                        # exactly-asserted same-scalar equalities inside G1?
                        self.ctx.check(N.cg1_subgroup_flags_enqueue(self.ctx.handle, b["pts"].ptr, L, n, self._exact_offsets,
                                                                    N_EXACT_POINTS, b["sgflags"].ptr))
                    self._decompress_collect(b, lo, hi)
                    tk["decompress_s"] = time.perf_counter() - t_dec   # staging + H2D + kernel + D2H of the batch's sub-batches
                    tk["chunks"].put((lo, hi))
                # the subgroup flags come back on THIS thread (a context takes one call at a time; the MSM stage runs on the other
                # lane): the front-end is already released, so only this lane's next batch waits for the side stream
                self.ctx.check(N.cg1_side_sync(self.ctx.handle))
                tk["sgflags"] = b["sgflags"].download(n * N_EXACT_POINTS)
            except BaseException as e:                          # surfaced in the consumer
                tk["chunks"].put(e)
            finally:
                tk["flags_done"].set()

        self._gpu_submit(gpu_stage)
        return tk

    def _front_end(self, tk: dict) -> None:
        if isinstance(tk["instances"], _Segments):            # (the host front-end takes contiguous bytes; coalescing is the device front-end's)
            tk["instances"], tk["proofs"] = tk["instances"].joined(), tk["proofs"].joined()
        return self._front_end_contiguous(tk)

    def _front_end_contiguous(self, tk: dict) -> None:
        """Stage 2 (caller's thread, all cores through the native pool): the front-end of each sub-batch as soon as the
        GPU has decoded it.  device_rows: it emits the row builder's input blocks (challenges + derived scalars), which go
        to the device with the host's reject codes; else the rows themselves, then point verdicts, early rejects and the
        summed CRS row on the host."""
        import time

        crs = self.crs
        L, C, K = crs.points_per_proof, crs.ncrs, self._rowin_scalars
        b, n = tk["slot"], tk["n"]
        host = b["host"]
        t0 = time.perf_counter()
        dev = self.device_rows
        tk["device_rows"] = dev
        prep = None if dev else Prepared(crs, n, False, host)
        tk["prep"] = prep
        hstat = (ctypes.c_int32 * n).from_address(host["hstat"].ptr)
        fe_first = tk.get("fe_first", False)
        if fe_first:
            todo = [(0, n)]                                   # the whole batch at once, before the GPU has decoded anything
        else:
            todo = tk["bounds"]
        for item in todo:
            if fe_first:
                r = item
            else:
                r = tk["chunks"].get()
            if isinstance(r, BaseException):
                tk["error"] = r
                tk["done"].set()
                raise r
            lo, hi = r
            args = (crs.handle, hi - lo, _addr(tk["instances"]) + lo * 4 * crs.ell * 48, _addr(tk["proofs"]) + lo * crs.proof_bytes,
                    _addr(tk["weights"]) + lo * N_WEIGHTS * 32, None if fe_first else host["decoded"].ptr + lo * 768, 768,
                    host["fe_wire"].ptr + lo * L * 48 if fe_first else host["wire"].ptr + lo * L * 48)
            if dev:
                rc = N.cg1_shuffle_prepare_inputs(*args, host["rowin"].ptr + lo * K * 32, host["hstat"].ptr + lo * 4, self.threads)
            else:
                rc = N.cg1_shuffle_prepare(*args, host["sc"].ptr + lo * L * 32, ctypes.addressof(prep.crs_scalars32) + lo * C * 32,
                                           ctypes.addressof(prep.status) + lo * 4, None, self.threads)
            if rc:
                tk["done"].set()
                raise N.NativeError(f"cg1_shuffle_prepare failed ({rc})")
        if fe_first:                                          # now the decoding must have finished (point verdicts, decoded points for the MSM)
            for _ in tk["bounds"]:
                r = tk["chunks"].get()
                if isinstance(r, BaseException):
                    tk["error"] = r
                    tk["done"].set()
                    raise r
        if dev:
            for i, s in enumerate(tk["pre_status"] or ()):
                if s:
                    hstat[i] = s
            # input blocks + host codes to the device on the MSM context's copy stream, from this thread
            self.ctx_msm.check(N.cg1_h2d_async(self.ctx_msm.handle, b["rowin"].ptr, host["rowin"].ptr, n * K * 32))
            self.ctx_msm.check(N.cg1_h2d_async(self.ctx_msm.handle, b["hstat"].ptr, host["hstat"].ptr, 4 * n))
            tk["front_end_s"] = time.perf_counter() - t0
            return
        self.ctx.check(N.cg1_shuffle_apply_point_status(prep.status, host["pstat"].buf, n, L, prep.scalars32, prep.crs_scalars32, C))
        for i, s in enumerate(tk["pre_status"] or ()):
            if s:
                prep.status[i] = s
                ctypes.memset(ctypes.addressof(prep.scalars32) + i * L * 32, 0, L * 32)
                ctypes.memset(ctypes.addressof(prep.crs_scalars32) + i * C * 32, 0, C * 32)
        crs_sum = ctypes.create_string_buffer(C * 32)
        self.ctx.check(N.cg1_shuffle_sum_crs_scalars(prep.crs_scalars32, prep.status, n, C, crs_sum))
        ctypes.memmove(host["sc"].ptr + n * L * 32, crs_sum, C * 32)
        # scalars to the device on the copy stream, from this thread, while the GPU thread is busy with other batches
        self.ctx_msm.check(N.cg1_h2d_async(self.ctx_msm.handle, b["sc"].ptr, host["sc"].ptr, (n * L + C) * 32))
        tk["front_end_s"] = time.perf_counter() - t0

    def _decide_flagged(self, tk: dict, i: int) -> bool:
        """MSM lane: proof i of the batch (it carries a point outside G1 in the same-scalar argument) on its own -- the weighted
        statement without the four same-scalar equalities (w1..w4 = 0) over its decoded points, which are still on the device,
        plus those four equalities exactly (cg1_shuffle_exact_same_scalar)."""
        crs, ctx = self.crs, self.ctx_msm
        L, C = crs.points_per_proof, crs.ncrs
        b, n = tk["slot"], tk["n"]
        ib, pb = 4 * crs.ell * 48, crs.proof_bytes
        inst = next(_runs(tk["instances"], ib, i, i + 1))[0]
        proof = next(_runs(tk["proofs"], pb, i, i + 1))[0]
        ok = ctypes.c_int(0)
        ctx.check(N.cg1_shuffle_exact_same_scalar(crs.handle, inst, proof, ctypes.byref(ok)))
        if not ok.value:
            return False
        w = bytearray(tk["weights"][i * N_WEIGHTS * 32: (i + 1) * N_WEIGHTS * 32])
        w[8 * 32: 12 * 32] = bytes(4 * 32)
        host = b["host"]
        sc, csc, st = ctypes.create_string_buffer(L * 32), ctypes.create_string_buffer(C * 32), (ctypes.c_int32 * 1)()
        rc = N.cg1_shuffle_prepare(crs.handle, 1, inst, proof, bytes(w), host["decoded"].ptr + i * 768, 768, host["wire"].ptr + i * L * 48,
                                   sc, csc, st, None, 1)
        if rc:
            raise N.NativeError(f"cg1_shuffle_prepare failed ({rc})")
        if st[0]:
            return False
        d_sc, d_csc = ctx.alloc(L * 32), ctx.alloc(C * 32)
        try:
            d_sc.upload(sc.raw); d_csc.upload(csc.raw)
            own = ctx.msm_device(b["pts"].ptr + i * L * 96, d_sc, L)
            shared = ctx.msm_device(b["pts"].ptr + n * L * 96, d_csc, C)
        finally:
            d_sc.free(); d_csc.free()
        tmp = ctypes.create_string_buffer(N.POINT_BYTES)
        N.cg1_add(tmp, own, shared)
        return bool(N.cg1_is_identity(tmp.raw))

    def _fe_lane(self, k: int, n: int):
        """Front-end lane k: its own context (stream), device front-end handle and aux staging, created on first use."""
        pair = self._fe[k]
        if pair is None or pair[2].nbytes < n * 19 * 32:
            if pair is not None:
                cx, fe, aux_h, aux_d = pair
                aux_h.free(); aux_d.free()
            else:
                cx = N.Context(self.ctx.device, cu_mask=range(self._cu_total - self.fe_cus, self._cu_total) if self.fe_cus else None)
                cx.set_param("fe_prio", self.fe_prio)
                fe = N.cg1_shuffle_fe_create(cx.handle, self.crs.ell, self.crs.lg, self.crs.affine96, self.crs.bytes)
                if not fe:
                    raise N.NativeError("cg1_shuffle_fe_create failed")
            pair = self._fe[k] = (cx, fe, N.PinnedBuffer(cx, n * 19 * 32), cx.alloc(n * 19 * 32))
        return pair

    def _front_end_device(self, tk: dict) -> None:
        """Stage 2 on the GPU (asynchronous): once the batch is decoded, one k_shuffle_front_end launch on the next front-end lane
        writes the row-input blocks and the front-end codes straight into the slot's device buffers."""
        import threading
        import time

        k = self._fe_next
        self._fe_next = (k + 1) % self.fe_lanes
        tk["device_rows"] = True
        tk["prep"] = None
        tk["fe_done"] = threading.Event()
        b, n = tk["slot"], tk["n"]

        def job():
            try:
                for _ in tk["bounds"]:
                    r = tk["chunks"].get()
                    if isinstance(r, BaseException):
                        raise r
                t0 = time.perf_counter()
                cx, fe, aux_h, aux_d = self._fe_lane(k, n)
                aux_rec = N.cg1_shuffle_fe_aux_bytes()
                for pa, a, cnt in _runs(tk["proofs"], self.crs.proof_bytes, 0, n):
                    cx.check(N.cg1_shuffle_gather_aux(self.crs.handle, cnt, pa, _addr(tk["weights"]) + a * N_WEIGHTS * 32, aux_h.ptr + a * aux_rec))
                cx.check(N.cg1_h2d(cx.handle, aux_d.ptr, aux_h.ptr, n * 19 * 32))
                cx.check(N.cg1_shuffle_fe_enqueue(fe, cx.handle, n, b["wire"].ptr, b["pts"].ptr, aux_d.ptr, b["rowin"].ptr, b["hstat"].ptr, 0))
                cx.check(N.cg1_stream_sync(cx.handle))
                pre = tk["pre_status"]
                if pre is not None and any(pre):                  # proofs rejected while packing (bad lengths): their codes win
                    hs = (ctypes.c_int32 * n).from_buffer_copy(b["hstat"].download(4 * n))
                    for i, s_ in enumerate(pre):
                        if s_:
                            hs[i] = s_
                    b["hstat"].upload(bytes(hs))
                tk["front_end_s"] = time.perf_counter() - t0
            except BaseException as e:
                tk["error"] = e
            finally:
                tk["fe_done"].set()

        self._gpu_submit(job, lane=2 + k)

    def _verify_stream_pipelines(self, batches, mode: str, rng):
        """verify_stream over `pipelines` child verifiers (same CRS, own contexts and threads) that take the batches in turn; verdicts
        are yielded in the order of the input."""
        import queue
        import threading
        from collections import deque

        if self._kids is None:
            self._release_lanes()                             # (lanes a single batch may have built on this verifier itself)
            dev = self.ctx.device
            self._kids = [ShuffleBatchVerifier(self.crs, N.Context(dev), threads=self.threads, chunk=self.chunk, device_rows=self.device_rows,
                                               blocking_sync=self.blocking_sync, device_front_end=True, fe_lanes=self.fe_lanes, fe_prio=self.fe_prio, pipelines=1,
                                               max_pinned_bytes=self.max_pinned_bytes, coalesce=0)      # (the stream is coalesced once, by this verifier: a child
                                                                                                       # waiting for a second batch to merge would starve its siblings' order)
                          for _ in range(self.pipelines)]
            for k in self._kids:
                k._own_ctx = True
        P = len(self._kids)
        depth = self.fe_lanes + 3
        in_q = [queue.Queue(maxsize=depth) for _ in range(P)]
        out_q = [queue.Queue() for _ in range(P)]

        def run(kid, qi, qo):
            try:
                for st in kid.verify_stream(iter(qi.get, None), mode, None):      # (weights were drawn by the feeding thread)
                    qo.put((st, dict(kid.last_stats)))
            except BaseException as e:                      # (reported to the consumer at this batch's turn)
                qo.put(e)
                while qi.get() is not None:                 # keep the feeder from blocking on a full queue
                    pass

        ths = [threading.Thread(target=run, args=(self._kids[i], in_q[i], out_q[i]), daemon=True) for i in range(P)]
        for t in ths:
            t.start()
        order = deque()

        def pop():
            r = out_q[order.popleft()].get()
            if isinstance(r, BaseException):
                raise r
            self.last_status, self.last_stats = r[0], r[1]
            return r[0]

        try:
            k = 0
            for batch in batches:
                if rng is not None:
                    # a caller's generator (tests, seeded runs) is used by THIS thread only, in batch order: reproducible weights, and
                    # no generator is shared between the pipelines' threads
                    batch = tuple(batch) + (None,) * (5 - len(batch))
                    if batch[4] is None:
                        batch = batch[:4] + (self.draw_weights(batch[2], rng),)
                in_q[k % P].put(batch)
                order.append(k % P)
                k += 1
                while len(order) > P * depth:
                    yield pop()
            for q in in_q:
                q.put(None)
            while order:
                yield pop()
        finally:
            for q, t in zip(in_q, ths):                      # (also when abandoned mid-way: drop what was not started, let the child drain)
                while t.is_alive():
                    try:
                        q.get_nowait()
                    except queue.Empty:
                        pass
                    try:
                        q.put(None, timeout=0.05)
                        break
                    except queue.Full:
                        continue
            for t in ths:
                t.join()

    def _verify_stream_device(self, batches, mode: str, rng):
        """verify_stream with the front-end on the GPU: up to fe_lanes + 2 batches in flight (decoding | front-end launches side by
        side | rows + merged MSM), verdicts yielded in order."""
        from collections import deque

        # several pipelines pay off on a STREAM of batches; one batch alone (is_valid_whisk_shuffle_proof, verify_many) runs on this
        # verifier's own lanes: no child verifiers, contexts or threads are made for it
        if self.pipelines > 1:
            if not (isinstance(batches, (list, tuple)) and len(batches) <= 1):
                yield from self._verify_stream_pipelines(batches, mode, rng)
                return
            if self._kids:                                    # the pipelines exist already: the batch rides the first one, in this thread
                kid = self._kids[0]
                for st in kid._verify_stream_device(batches, mode, rng):
                    self.last_status, self.last_stats = st, dict(kid.last_stats)
                    yield st
                return

        inflight = deque()
        depth = self.fe_lanes + 2
        try:
            for batch in batches:
                tk = self._begin(batch, mode, rng, prefetched=True)
                depth = min(depth, max(1, self._nslots - 2))       # (a pinned-memory budget may have shrunk the slot rotation: fewer batches in flight)
                self._front_end_device(tk)
                self._enqueue_msm(tk)
                inflight.append(tk)
                while len(inflight) > depth:
                    yield self._finish(inflight.popleft())
            while inflight:
                yield self._finish(inflight.popleft())
        finally:
            left = [t for t in inflight if not t["done"].is_set()]
            for t in left:
                t["done"].wait()

    def _enqueue_msm(self, tk: dict) -> None:
        """Stage 3 (asynchronous, GPU thread): scalars H2D, the merged MSM, and -- if it is not the identity, or in mode
        "independent" -- the per-proof MSMs that name the invalid proofs."""
        import time

        crs, ctx = self.crs, self.ctx_msm
        L, C = crs.points_per_proof, crs.ncrs
        b, n, prep = tk["slot"], tk["n"], tk["prep"]

        def gpu_stage():
            try:
                if tk.get("fe_done") is not None:                 # device front-end: its launch wrote the blocks and codes into the slot
                    tk["fe_done"].wait()
                    if tk["error"] is not None:
                        raise tk["error"]
                t0 = time.perf_counter()
                ctx.check(N.cg1_copy_fence(ctx.handle))          # the scalars / input blocks queued by _front_end
                if tk["device_rows"]:
                    # the rows are expanded on the device, the point verdicts folded in there, the CRS rows summed behind them
                    ctx.check(N.cg1_shuffle_rows_device(ctx.handle, crs.ell, crs.lg, n, b["rowin"].ptr, b["hstat"].ptr, b["pstat"].ptr,
                                                        b["sc"].ptr, b["csrows"].ptr, b["dstat"].ptr))
                    ctx.check(N.cg1_stream_sync(ctx.handle))
                    status = list((ctypes.c_int32 * n).from_buffer_copy(b["dstat"].download(4 * n)))
                else:
                    status = [int(prep.status[i]) for i in range(n)]
                live = [i for i in range(n) if status[i] == 0]
                merged_ok = None
                if live and tk["mode"] == "merged":
                    merged_ok = bool(N.cg1_is_identity(ctx.msm_device(b["pts"], b["sc"], n * L + C)))
                t1 = time.perf_counter()
                if live and not merged_ok:
                    # independent: P_i over the proof's own points, Q_i over the CRS points; valid iff P_i + Q_i = 0
                    own = ctx.msm_batched_device(b["pts"], b["sc"], [i * L for i in range(n + 1)])
                    rep_pts = ctx.alloc(n * C * 96)
                    rep_pts.upload(crs.affine96 * n)
                    if tk["device_rows"]:
                        shared = ctx.msm_batched_device(rep_pts, b["csrows"], [i * C for i in range(n + 1)])
                    else:
                        rep_sc = ctx.alloc(n * C * 32)
                        ctx.check(N.cg1_h2d(ctx.handle, rep_sc.ptr, prep.crs_scalars32, n * C * 32))
                        shared = ctx.msm_batched_device(rep_pts, rep_sc, [i * C for i in range(n + 1)])
                        rep_sc.free()
                    rep_pts.free()
                    tmp = ctypes.create_string_buffer(N.POINT_BYTES)
                    for i in live:
                        N.cg1_add(tmp, own[i], shared[i])
                        if not N.cg1_is_identity(tmp.raw):
                            status[i] = REJECT_EQUATION
                # Proofs carrying a point outside G1 in the same-scalar equalities.  The reference asserts those four equalities
                # EXACTLY (same_scalar.py:101-108), with its scalars as integers in [0, r): a random weight reduced mod r is blind
                # to a torsion defect with probability 1/3, and -- the other direction -- torsion components that cancel in the
                # exact equalities (the reference accepts) do NOT cancel once T_1, U_1, cm_A, cm_B carry different weights.  So a
                # flagged proof is decided apart from the batch, whatever the weighted passes above said about it: its statement
                # again with the same-scalar weights w1..w4 (rho[8..11]) set to ZERO -- every check the reference itself runs
                # through its randomised MSMAccumulator, where A' = A + T_1 + U_1 is formed from the decoded points as the
                # reference forms it -- AND the four equalities evaluated without weights on the host.
                # (None in honest traffic.  The flags were brought back by the decompression lane: tk["sgflags"].)
                tk["flags_done"].wait()
                flags = tk["sgflags"]
                n_exact = 0
                if flags is not None and any(flags):
                    was_live = set(live)
                    for i in range(n):
                        if i in was_live and any(flags[i * N_EXACT_POINTS: (i + 1) * N_EXACT_POINTS]):
                            n_exact += 1
                            status[i] = 0 if self._decide_flagged(tk, i) else REJECT_EQUATION
                tk["status"] = status
                tk["stats"] = {"merged_msm_s": t1 - t0, "exact_checks": n_exact, "independent_s": time.perf_counter() - t1, "merged_ok": merged_ok,
                               "front_end_s": tk.get("front_end_s", 0.0), "decompress_s": tk.get("decompress_s", 0.0), "n": n, "points": n * L + C, "pipelined": len(tk["bounds"]) > 1}
            except BaseException as e:
                tk["error"] = e
            finally:
                tk["done"].set()

        self._gpu_submit(gpu_stage, lane=1)

    def _finish(self, tk: dict) -> List[int]:
        import time

        tk["done"].wait()
        if tk["error"] is not None:
            raise tk["error"]
        self.last_status = tk["status"]
        self.last_stats = dict(tk["stats"])
        self.last_stats["total_s"] = time.perf_counter() - tk["t0"]
        return tk["status"]

    def verify_stream(self, batches, mode: str = "merged", rng=None):
        """Verify a sequence of batches, three stages overlapped across batches: while the host front-end works on batch k,
        the GPU finishes the MSM of batch k-1 and already decompresses batch k+1.
        `batches` yields (instances, proofs, n[, pre_status[, weights]]); yields one status list (0 = valid) per batch."""
        if self.device_front_end:
            if self._host_twin is not None:            # (its lanes would keep hardware queues the pipelines need)
                self._host_twin.close()
                self._host_twin = None
            if self.coalesce and not (isinstance(batches, (list, tuple)) and len(batches) <= 1):
                yield from self._verify_stream_coalesced(batches, mode, rng)
                return
            yield from self._verify_stream_device(batches, mode, rng)
            return
        it = iter(batches)
        cur = next(it, None)
        if cur is None:
            return
        tk = self._begin(cur, mode, rng)
        pending = tk_next = None
        try:
            while tk is not None:
                nxt = next(it, None)
                tk_next = self._begin(nxt, mode, rng, prefetched=True) if nxt is not None else None    # decompression of the next batch
                self._front_end(tk)
                self._enqueue_msm(tk)
                if pending is not None:
                    done, pending = pending, None
                    yield self._finish(done)
                pending, tk, tk_next = tk, tk_next, None
            if pending is not None:
                done, pending = pending, None
                yield self._finish(done)
        finally:
            # abandoned mid-way (an error, or the consumer stopped iterating): release the buffer slots of the batches
            # still in flight once the GPU thread has drained what was queued for them
            left = [t for t in (pending, tk, tk_next) if t is not None and not t["done"].is_set()]
            if left:
                self._gpu_submit(lambda: self._gpu_submit(lambda: [t["done"].set() for t in left], lane=1))

    def _verify_stream_coalesced(self, batches, mode: str, rng):
        """verify_stream for a process with few hardware queues (the runtime's default is 4): a front-end launch takes ~12 ms whatever it
        carries and only one or two run side by side there, so consecutive batches of the stream travel together -- up to `coalesce` proofs
        per internal batch (one front-end launch, one decoding pass, one merged MSM for all of them) -- and every caller's batch still gets
        its own status list, in order.  Batches that bring their own pre_status / weights, and batches already that large, go through
        as they are.  (profiles/r05_verify_queues_ab.txt: 4 queues, 1 024 proofs per batch: 7.8e4 proofs/s alone, ~1.5e5 in fours.)"""
        from collections import deque

        limit = int(self.coalesce)
        groups = deque()                                        # per internal batch: the sizes of the callers' batches inside it

        ib, pb = 4 * self.crs.ell * 48, self.crs.proof_bytes

        def together(pend, size):
            """the callers' batches as one internal batch, NOT copied together: the stages gather from the segments where they lie"""
            if len(pend) == 1:
                return pend[0]
            for x in pend:
                assert len(x[0]) == x[2] * ib and len(x[1]) == x[2] * pb
            return (_Segments([(x[0], x[2]) for x in pend], ib), _Segments([(x[1], x[2]) for x in pend], pb), size)

        def merged():
            pend, size = [], 0
            for b in batches:
                n = b[2]
                plain = len(b) == 3
                if not plain or n >= limit or (pend and size + n > limit):
                    if pend:
                        groups.append([x[2] for x in pend])
                        yield together(pend, size)
                        pend, size = [], 0
                    if not plain or n >= limit:
                        groups.append([n])
                        yield b
                        continue
                pend.append(b)
                size += n
                if size >= limit:
                    groups.append([x[2] for x in pend])
                    yield together(pend, size)
                    pend, size = [], 0
            if pend:
                groups.append([x[2] for x in pend])
                yield together(pend, size)

        for st in self._verify_stream_device(merged(), mode, rng):
            sizes = groups.popleft()
            lo = 0
            for n in sizes:
                yield st[lo: lo + n]
                lo += n

    def verify_packed(self, instances: bytes, proofs: bytes, n: int, mode: str = "merged", rng=None, weights=None,
                      pre_status: Optional[Sequence[int]] = None) -> List[int]:
        """One batch in wire form: `instances` = n x (vec_R | vec_S | vec_T | vec_U) encodings, `proofs` = n x
        crs.proof_bytes.  Returns the per-proof status (0 = valid, else a reject code of REJECT_NAMES)."""
        if n == 0:
            return []
        if n > 2 * self.SPLIT:
            # a very large call (BASELINE config 5 hands over 16 384 proofs) goes through the stream in pieces of SPLIT proofs: buffers
            # stay at the size of a piece and the stages of consecutive pieces overlap (profiles/r03_verify_batchsize.txt: 2048 per
            # piece is where proofs/s levels off)
            ib, pb = 4 * self.crs.ell * 48, self.crs.proof_bytes
            inst, prf = memoryview(instances), memoryview(proofs)
            w = memoryview(weights) if weights is not None else None

            def pieces():
                for a in range(0, n, self.SPLIT):
                    m = min(self.SPLIT, n - a)
                    yield (bytes(inst[a * ib: (a + m) * ib]), bytes(prf[a * pb: (a + m) * pb]), m,
                           None if pre_status is None else list(pre_status[a: a + m]),
                           None if w is None else bytes(w[a * N_WEIGHTS * 32: (a + m) * N_WEIGHTS * 32]))

            out: List[int] = []
            total: dict = {}
            for st in self.verify_stream(pieces(), mode=mode, rng=rng):
                out += st
                for k, v in (self.last_stats or {}).items():             # the call's statistics: summed over its pieces
                    if isinstance(v, bool) or not isinstance(v, (int, float)):
                        total[k] = v
                    else:
                        total[k] = total.get(k, 0) + v
            self.last_status, self.last_stats = out, total
            return out
        if self.device_front_end and self._fe_by_default:
            # ONE batch on its own is a matter of latency, not of throughput: the device front-end answers in ~12.5 ms + 30 us per proof
            # whatever the batch (a transcript is ~800 dependent Keccak passes), the host front-end in ~1.4 ms + 1 ms per proof over the
            # host's threads (profiles/r04_single_proof_latency.txt: one proof 1.4 against 12.7 ms, 256 proofs 5.2 against 14.2 on 16
            # threads).  IsValidWhiskShuffleProof is a batch of one.  Streams of batches (verify_stream) keep the device front-end.
            threads = self.threads or int(N.cg1_shuffle_default_threads())
            host_ms = 1.4 + 1.05 * n / max(1, min(threads, n))
            if host_ms < 12.5 + 0.03 * n:
                if self._host_twin is None:
                    self._host_twin = ShuffleBatchVerifier(self.crs, self._ctx if not getattr(self, "_own_ctx", False) else None, threads=self.threads,
                                                           chunk=self.chunk, device_rows=self.device_rows, blocking_sync=self.blocking_sync, device_front_end=False)
                tw = self._host_twin
                out = tw.verify_packed(instances, proofs, n, mode=mode, rng=rng, weights=weights, pre_status=pre_status)
                self.last_status, self.last_stats = tw.last_status, tw.last_stats
                return out
        return next(self.verify_stream([(instances, proofs, n, pre_status, weights)], mode=mode, rng=rng))

    def verify_many(self, items, mode: str = "merged", rng=None) -> List[bool]:
        """[IsValidWhiskShuffleProof(crs, pre, post, proof) for (pre, post, proof) in items] (whisk_interface.py:72-87)."""
        items = list(items)
        if not items:
            return []
        inst, proofs, pre_status = self.pack(items)
        status = self.verify_packed(inst, proofs, len(items), mode=mode, rng=rng, pre_status=pre_status)
        return [s == 0 for s in status]


_verifier_cache: dict = {}


def verifier_for(crs, ctx=None) -> ShuffleBatchVerifier:
    """One ShuffleBatchVerifier per (CRS bytes, context), kept across calls: creating one decodes the CRS and builds the
    fixed-base tables of G_sum / H_sum (tens of milliseconds)."""
    if isinstance(crs, ShuffleBatchVerifier):
        return crs
    data = crs.bytes if isinstance(crs, ShuffleCrs) else (bytes(crs.to_bytes()) if hasattr(crs, "to_bytes") else bytes(crs))
    key = (data, id(ctx))
    v = _verifier_cache.get(key)
    if v is None:
        if len(_verifier_cache) >= 4:
            _verifier_cache.pop(next(iter(_verifier_cache))).close()
        v = _verifier_cache[key] = ShuffleBatchVerifier(crs if isinstance(crs, ShuffleCrs) else data, ctx)
    return v


def is_valid_whisk_shuffle_proof(crs, pre_shuffle_trackers, post_shuffle_trackers, whisk_shuffle_proof_bytes, ctx=None) -> bool:
    """Drop-in for IsValidWhiskShuffleProof (whisk_interface.py:72-87) -- a batch of one."""
    return verifier_for(crs, ctx).verify_many([(pre_shuffle_trackers, post_shuffle_trackers, whisk_shuffle_proof_bytes)])[0]


def are_valid_whisk_shuffle_proofs(crs, items, ctx=None) -> List[bool]:
    """[IsValidWhiskShuffleProof(crs, pre, post, proof) for (pre, post, proof) in items] in one batch."""
    return verifier_for(crs, ctx).verify_many(items)


class OpeningBatchVerifier:
    """Many `IsValidWhiskOpeningProof(tracker, k_commitment, proof)` calls (whisk_interface.py:147-169) as one GPU MSM of
    5 n + 1 terms: both equalities of every proof (opening.py:74-77) under fresh random weights, points decompressed on
    the GPU.  If the merged check fails, one 5-term MSM per proof (plus its generator term on the host) names the culprits.

    `device_front_end` (default): the transcripts and the scalars of the merged check are produced on the GPU too
    (`cg1_opening_prepare_device`, csrc/kernels_opening.h): the host only hands the wire bytes over.  False = the host front-end
    (`cg1_opening_prepare` on the native worker pool); both give the same verdicts and status codes (tests/test_opening_batch.py)."""

    PROOF_BYTES = 128                       # A | B | s   (opening.py:94-99)
    CULPRIT_SLICE = 32768                   # per-proof MSMs of one regime-B call when the merged check fails
    SMALL_EXACT = 4                         # up to this many proofs are checked one by one on the host (cg1_opening_exact_status: ~0.4 ms each;
                                            # the GPU path is ~2.4 ms of dependent launches whatever the batch: IsValidWhiskOpeningProof is a batch of one)

    def __init__(self, ctx: Optional["N.Context"] = None, device_front_end: bool = True):
        self._ctx = ctx
        self.device_front_end = bool(device_front_end)
        g = ctypes.create_string_buffer(N.POINT_BYTES)
        N.cg1_generator(g)
        aff = ctypes.create_string_buffer(96)
        N.cg1_to_affine96(aff, g.raw)
        self._g_blob, self._g96 = g.raw, aff.raw
        self.last_status: List[int] = []
        self._dev = None                      # (capacity, d_pts, d_sc) kept between calls

    @property
    def ctx(self) -> "N.Context":
        if self._ctx is None:
            self._ctx = N.default_context()
        return self._ctx

    @staticmethod
    def _weights(n, rng, seed=None):
        """rng given (tests): 2 n weights from it.  Otherwise both front-ends derive them from `seed` (32 fresh bytes from the OS per batch):
        SHAKE256(seed || le64(i)) -> two 128-bit weights per proof (cg1_opening_weights_from_seed; on the device inside k_opening_scalars)."""
        if rng is not None:
            return b"".join(rng.randint(1, FR_MODULUS - 1).to_bytes(32, "little") for _ in range(2 * n))
        out = ctypes.create_string_buffer(max(1, 64 * n))
        rc = N.cg1_opening_weights_from_seed(seed, 0, n, out)
        if rc:
            raise N.NativeError(f"cg1_opening_weights_from_seed failed ({rc})")
        return out.raw[: 64 * n]

    def _pack(self, items):
        """items: (tracker, k_commitment, proof_bytes); tracker = WhiskTracker-like or (r_G, k_r_G) -> packed wire arrays + length verdicts"""
        items = list(items)
        n = len(items)
        trk = [it[0] for it in items]
        rs = [t.r_G if hasattr(t, "r_G") else t[0] for t in trk]
        krs = [t.k_r_G if hasattr(t, "k_r_G") else t[1] for t in trk]
        ks = [it[1] for it in items]
        ps = [it[2] for it in items]
        try:
            uniform = ({len(x) for x in rs} | {len(x) for x in krs} | {len(x) for x in ks}) <= {48} and {len(x) for x in ps} <= {self.PROOF_BYTES}
        except TypeError:
            uniform = False
        if uniform:                                   # the common case: everything well-formed, packed without a per-item loop
            return n, b"".join(x for pair in zip(rs, krs) for x in pair), b"".join(ks), b"".join(ps), None
        tr, kc, pf, pre = [], [], [], []
        for r, kr, k, p in zip(rs, krs, ks, ps):
            r, kr, k, p = bytes(r), bytes(kr), bytes(k), bytes(p)
            ok = len(r) == len(kr) == len(k) == 48 and len(p) >= self.PROOF_BYTES       # BufReader ignores trailing bytes
            tr.append(r + kr if ok else bytes(96)); kc.append(k if ok else bytes(48)); pf.append(p[:128] if ok else bytes(128))
            pre.append(0 if ok else REJECT_LENGTH)
        return n, b"".join(tr), b"".join(kc), b"".join(pf), pre

    def prepare(self, items, rng=None):
        """Host front-end only (no GPU): points in MSM order, scalars, generator scalars and status per proof."""
        n, trackers, kcs, pfs, pre = self._pack(items)
        return self._prepare_host(n, trackers, kcs, pfs, pre, self._weights(n, rng, secrets.token_bytes(32)))

    def _prepare_host(self, n, trackers, kcs, pfs, pre, weights):
        out = {"n": n, "points48": ctypes.create_string_buffer(max(1, 240 * n)), "scalars32": ctypes.create_string_buffer(max(1, 160 * n)),
               "g_scalars32": ctypes.create_string_buffer(max(1, 32 * n)), "status": (ctypes.c_int32 * max(1, n))()}
        rc = N.cg1_opening_prepare(n, trackers, kcs, pfs, weights, out["points48"], out["scalars32"],
                                   out["g_scalars32"], out["status"])
        if rc:
            raise N.NativeError(f"cg1_opening_prepare failed ({rc})")
        for i, s in enumerate(pre or ()):
            if s:
                out["status"][i] = s
                ctypes.memset(ctypes.addressof(out["scalars32"]) + 160 * i, 0, 160)
                ctypes.memset(ctypes.addressof(out["g_scalars32"]) + 32 * i, 0, 32)
        return out

    def _buffers(self, n):
        """Device buffers for n proofs (grown geometrically; the old pair is freed first).  One verifier = one caller at a time: its buffers
        and `last_status` are per-object state (use one OpeningBatchVerifier per thread)."""
        if self._dev is None or self._dev[0] < n:
            ctx = self.ctx
            if self._dev is not None:
                self._dev[1].free()
                self._dev[2].free()
                self._dev = None
            cap = max(n, 1024, 2 * (self._dev_cap_seen if hasattr(self, "_dev_cap_seen") else 0))
            self._dev_cap_seen = cap
            self._dev = (cap, ctx.alloc(96 * (5 * cap + 1)), ctx.alloc(32 * (5 * cap + 1)))
        return self._dev[1], self._dev[2]

    def verify_many(self, items, rng=None, _seed: Optional[bytes] = None) -> List[bool]:
        """One verdict per (tracker, k_commitment, proof).  The batch weights come from 32 fresh bytes of the OS per call; `_seed` replaces
        them for TESTS that compare two front-ends on the same weights -- a seed that is reused or known to the prover makes every weight
        predictable, and a forged batch can then pass: never pass it in production."""
        n, trackers, kcs, pfs, pre = self._pack(items)
        return self._verify(n, trackers, kcs, pfs, pre, rng, _seed)

    def verify_packed(self, trackers96: bytes, k_commitments48: bytes, proofs128: bytes, rng=None, _seed: Optional[bytes] = None) -> List[bool]:
        """The same verdicts for n proofs already laid out back to back: n x (r_G | k_r_G), n x k_commitment, n x (A | B | s)."""
        n = len(proofs128) // self.PROOF_BYTES
        if len(proofs128) != 128 * n or len(trackers96) != 96 * n or len(k_commitments48) != 48 * n:
            raise ValueError("verify_packed: expected n x 96, n x 48 and n x 128 bytes")
        return self._verify(n, trackers96, k_commitments48, proofs128, None, rng, _seed)

    def _verify(self, n, trackers, kcs, pfs, pre, rng, seed=None) -> List[bool]:
        if n == 0:
            self.last_status = []
            return []
        if n <= self.SMALL_EXACT:
            # the reference's own two equalities (opening.py:73-74), asserted exactly, proof by proof: no weights, no GPU round trips
            st = ctypes.c_int(0)
            status = []
            for i in range(n):
                rc = N.cg1_opening_exact_status(trackers[96 * i: 96 * i + 96], kcs[48 * i: 48 * i + 48], pfs[128 * i: 128 * i + 128], ctypes.byref(st))
                if rc:
                    raise N.NativeError(f"cg1_opening_exact_status failed ({rc})")
                status.append((pre[i] if pre else 0) or int(st.value))
            self.last_status = status
            return [s == 0 for s in status]
        ctx = self.ctx
        if seed is None:
            seed = secrets.token_bytes(32)
        elif len(seed) != 32:
            raise ValueError("seed: 32 bytes")
        d_pts, d_sc = self._buffers(n)
        # Both equalities of an opening proof are asserted EXACTLY by the reference (opening.py:74-77) on points it decodes
        # unchecked: weighting them randomly is sound only inside G1, so every point is decoded with the subgroup test.  A
        # proof with a point outside G1 leaves the batch (its scalars are zeroed) and is decided by the exact host check.
        if self.device_front_end:
            st_arr = (ctypes.c_int32 * n)()
            ps_buf = ctypes.create_string_buffer(5 * n)
            g_scalars = ctypes.create_string_buffer(32 * n)
            ctx.check(N.cg1_opening_prepare_device(ctx.handle, n, trackers, kcs, pfs, self._weights(n, rng) if rng is not None else None, seed,
                                                   d_pts.ptr, d_sc.ptr, st_arr, ps_buf, g_scalars))
            pstat = ps_buf.raw
            status = memoryview(st_arr).cast("B").cast("i").tolist()
        else:
            prep = self._prepare_host(n, trackers, kcs, pfs, pre, self._weights(n, rng, seed))
            d_wire, d_stat = ctx.alloc(240 * n), ctx.alloc(5 * n)
            d_wire.upload(prep["points48"].raw[: 240 * n])
            ctx.check(N.cg1_batch_decompress_device(ctx.handle, d_wire.ptr, d_pts.ptr, d_stat.ptr, 5 * n, 1))
            d_pts.upload(self._g96, 96 * 5 * n)
            pstat = d_stat.download(5 * n)
            ctx.check(N.cg1_shuffle_apply_point_status(prep["status"], pstat, n, 5, prep["scalars32"], prep["g_scalars32"], 1))
            g_sum = ctypes.create_string_buffer(32)
            ctx.check(N.cg1_shuffle_sum_crs_scalars(prep["g_scalars32"], prep["status"], n, 1, g_sum))
            d_sc.upload(prep["scalars32"].raw[: 160 * n] + g_sum.raw)
            g_scalars = prep["g_scalars32"]
            status = [int(prep["status"][i]) for i in range(n)]
        if pre:
            status = [p or s for p, s in zip(pre, status)]
        if N.ERR_NOT_IN_SUBGROUP in pstat:
            outside = [i for i in range(n) if N.ERR_NOT_IN_SUBGROUP in pstat[5 * i: 5 * i + 5]
                       and not any(x not in (0, N.ERR_NOT_IN_SUBGROUP) for x in pstat[5 * i: 5 * i + 5])]
        else:
            outside = []
        if any(s == 0 for s in status) and not N.cg1_is_identity(ctx.msm_device(d_pts, d_sc, 5 * n + 1)):
            # the merged check failed: one 5-term MSM per proof names the culprits (a regime-B call takes at most 65 535 MSMs: slices)
            tmp = ctypes.create_string_buffer(N.POINT_BYTES)
            gs_raw = g_scalars.raw
            for lo in range(0, n, self.CULPRIT_SLICE):
                cnt = min(self.CULPRIT_SLICE, n - lo)
                own = ctx.msm_batched_device(d_pts.ptr + 96 * 5 * lo, d_sc.ptr + 32 * 5 * lo, [5 * i for i in range(cnt + 1)])
                for k in range(cnt):
                    i = lo + k
                    if status[i]:
                        continue
                    N.cg1_mul(tmp, self._g_blob, gs_raw[32 * i: 32 * i + 32])
                    N.cg1_add(tmp, tmp.raw, own[k])
                    if not N.cg1_is_identity(tmp.raw):
                        status[i] = REJECT_EQUATION
        ok = ctypes.c_int(0)
        for i in outside:
            if status[i] == 2:                       # rejected only for the subgroup flag: the exact equalities decide
                ctx.check(N.cg1_opening_exact(trackers[96 * i: 96 * i + 96], kcs[48 * i: 48 * i + 48], pfs[128 * i: 128 * i + 128], ctypes.byref(ok)))
                status[i] = 0 if ok.value else REJECT_EQUATION
        self.last_status = status
        return [s == 0 for s in status]


def is_valid_whisk_opening_proof(tracker, k_commitment, tracker_proof, ctx=None) -> bool:
    """Drop-in for IsValidWhiskOpeningProof (whisk_interface.py:147-160) -- a batch of one."""
    return OpeningBatchVerifier(ctx).verify_many([(tracker, k_commitment, tracker_proof)])[0]
