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
import secrets
from typing import List, Optional, Sequence, Tuple

from . import _native as N

FR_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # util.py:7
N_BLINDERS = 4                                                                   # curdleproofs.py:26
N_WEIGHTS = 12

REJECT_NAMES = {0: "prepared", 1: "bad scalar encoding", 2: "bad point encoding", 3: "vec_T[0] is infinity", 4: "bad weight",
                5: "bad length", 6: "verification equation failed"}
REJECT_LENGTH, REJECT_EQUATION = 5, 6


def _addr(b) -> int:
    """Address of the first byte of a bytes object / ctypes buffer (to pass sub-ranges of a batch to native code)."""
    if isinstance(b, bytes):
        return ctypes.cast(ctypes.c_char_p(b), ctypes.c_void_p).value
    return ctypes.addressof(b)


def _tracker_bytes(trackers) -> Tuple[bytes, bytes]:
    """Sequence of WhiskTracker-likes (r_G, k_r_G attributes; whisk_interface.py:24-30) or (r_G, k_r_G) pairs."""
    rs, ks = [], []
    for t in trackers:
        r, k = (t.r_G, t.k_r_G) if hasattr(t, "r_G") else t
        rs.append(bytes(r))
        ks.append(bytes(k))
    return b"".join(rs), b"".join(ks)


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
        self.crs_scalars32 = ctypes.create_string_buffer(max(1, n * C * 32))
        self.status = (ctypes.c_int32 * max(1, n))()
        self.challenges = ctypes.create_string_buffer(n * crs.challenges_per_proof * 32) if want_challenges and n else None


class ShuffleBatchVerifier:
    def __init__(self, crs, ctx: Optional["N.Context"] = None, threads: int = 0, chunk: int = 256):
        self.crs = crs if isinstance(crs, ShuffleCrs) else ShuffleCrs(crs)
        self._ctx = ctx
        self.threads = threads
        self.chunk = chunk                  # sub-batch of the decompress / front-end pipeline
        self._gpu_thread = None
        self._gpu_jobs = None
        self._bufs = {}
        self.last_stats = {}

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
            ok = len(pr) == len(pk) == len(qr) == len(qk) == 48 * ell and len(proof) >= pb
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
            raw[31::32] = bytes(b & 0x3F for b in raw[31::32])          # < 2^254 < r: canonical without rejection
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
            self._ctx = N.default_context()
        return self._ctx

    def _device_buffers(self, n: int):
        crs = self.crs
        L, C = crs.points_per_proof, crs.ncrs
        b = self._bufs.get(n)
        if b is None:
            self._bufs.clear()
            ctx = self.ctx
            b = {
                "wire": ctx.alloc(n * L * 48),
                "pts": ctx.alloc((n * L + C) * 96),          # own points of all proofs, then the CRS points
                "pstat": ctx.alloc(n * L),
                "sc": ctx.alloc((n * L + C) * 32),
                "host": {                                    # page-locked staging
                    "wire": N.PinnedBuffer(ctx, n * L * 48),
                    "sc": N.PinnedBuffer(ctx, (n * L + C) * 32),
                    "pstat": N.PinnedBuffer(ctx, n * L),
                    "decoded": N.PinnedBuffer(ctx, n * 768),
                },
            }
            b["pts"].upload(crs.affine96, n * L * 96)
            self._bufs[n] = b
        return b

    def decompress_on_gpu(self, instances: bytes, proofs: bytes, n: int, lo: int = 0, hi: Optional[int] = None):
        """Gather the own points of proofs [lo, hi) of an n-proof batch, decompress them on the GPU (they stay there for
        the MSM) and bring back the per-point verdicts + the 8-point window the host front-end wants."""
        crs, ctx = self.crs, self.ctx
        L = crs.points_per_proof
        hi = n if hi is None else hi
        m = hi - lo
        b = self._device_buffers(n)
        h = b["host"]
        wire = h["wire"].ptr + lo * L * 48
        ctx.check(N.cg1_shuffle_gather_points(crs.handle, m, _addr(instances) + lo * 4 * crs.ell * 48,
                                              _addr(proofs) + lo * crs.proof_bytes, wire))
        ctx.check(N.cg1_h2d(ctx.handle, b["wire"].ptr + lo * L * 48, wire, m * L * 48))
        ctx.check(N.cg1_batch_decompress_device(ctx.handle, b["wire"].ptr + lo * L * 48, b["pts"].ptr + lo * L * 96,
                                                b["pstat"].ptr + lo * L, m * L, 0))
        ctx.check(N.cg1_d2h(ctx.handle, h["pstat"].ptr + lo * L, b["pstat"].ptr + lo * L, m * L))
        ctx.check(N.cg1_d2h_2d(ctx.handle, h["decoded"].ptr + lo * 768, 768, b["pts"].ptr + (lo * L + 4 * crs.ell + 1) * 96, L * 96, 768, m))
        return h["pstat"].buf, h["decoded"].buf

    def check_prepared(self, prep: Prepared, mode: str = "merged", points_on_device: bool = False) -> List[int]:
        """Run the group arithmetic for a Prepared batch.  Returns the final per-proof status (0 = valid)."""
        import time

        crs, ctx, n = self.crs, self.ctx, prep.n
        L, C = crs.points_per_proof, crs.ncrs
        if n == 0:
            return []
        t0 = time.perf_counter()
        b = self._device_buffers(n)
        if not points_on_device:
            ctx.check(N.cg1_h2d(ctx.handle, b["wire"].ptr, prep.points48, n * L * 48))
            ctx.check(N.cg1_batch_decompress_device(ctx.handle, b["wire"].ptr, b["pts"].ptr, b["pstat"].ptr, n * L, 0))
            pstat = b["pstat"].download(n * L)
            ctx.check(N.cg1_shuffle_apply_point_status(prep.status, pstat, n, L, prep.scalars32, prep.crs_scalars32, C))
        t1 = time.perf_counter()
        status = [int(prep.status[i]) for i in range(n)]
        live = [i for i in range(n) if status[i] == 0]
        merged_ok = None
        if live and mode == "merged":
            crs_sum = ctypes.create_string_buffer(C * 32)
            ctx.check(N.cg1_shuffle_sum_crs_scalars(prep.crs_scalars32, prep.status, n, C, crs_sum))
            if len(prep.scalars32) >= (n * L + C) * 32:          # staging buffer: CRS row right behind, one copy
                ctypes.memmove(ctypes.addressof(prep.scalars32) + n * L * 32, crs_sum, C * 32)
                ctx.check(N.cg1_h2d(ctx.handle, b["sc"].ptr, prep.scalars32, (n * L + C) * 32))
            else:
                ctx.check(N.cg1_h2d(ctx.handle, b["sc"].ptr, prep.scalars32, n * L * 32))
                ctx.check(N.cg1_h2d(ctx.handle, b["sc"].ptr + n * L * 32, crs_sum, C * 32))
            blob = ctx.msm_device(b["pts"], b["sc"], n * L + C)
            merged_ok = bool(N.cg1_is_identity(blob))
        t2 = time.perf_counter()
        if live and not merged_ok:
            # independent: P_i over the proof's own points, Q_i over the CRS points; valid iff P_i + Q_i = 0
            ctx.check(N.cg1_h2d(ctx.handle, b["sc"].ptr, prep.scalars32, n * L * 32))
            own = ctx.msm_batched_device(b["pts"], b["sc"], [i * L for i in range(n + 1)])
            rep_pts = ctx.alloc(n * C * 96)
            rep_sc = ctx.alloc(n * C * 32)
            rep_pts.upload(crs.affine96 * n)
            ctx.check(N.cg1_h2d(ctx.handle, rep_sc.ptr, prep.crs_scalars32, n * C * 32))
            shared = ctx.msm_batched_device(rep_pts, rep_sc, [i * C for i in range(n + 1)])
            tmp = ctypes.create_string_buffer(N.POINT_BYTES)
            for i in live:
                N.cg1_add(tmp, own[i], shared[i])
                if not N.cg1_is_identity(tmp.raw):
                    status[i] = REJECT_EQUATION
        t3 = time.perf_counter()
        self.last_stats.update({"decompress_s": t1 - t0, "merged_msm_s": t2 - t1, "independent_s": t3 - t2, "merged_ok": merged_ok,
                                "n": n, "points": n * L + C})
        return status

    def verify_many(self, items, mode: str = "merged", rng=None) -> List[bool]:
        """[IsValidWhiskShuffleProof(crs, pre, post, proof) for (pre, post, proof) in items] (whisk_interface.py:72-87)."""
        import time

        items = list(items)
        if not items:
            return []
        t0 = time.perf_counter()
        inst, proofs, pre_status = self.pack(items)
        t1 = time.perf_counter()
        status = self.verify_packed(inst, proofs, len(items), mode=mode, rng=rng, pre_status=pre_status)
        self.last_stats["pack_s"] = t1 - t0
        self.last_stats["total_s"] = time.perf_counter() - t0
        return [s == 0 for s in status]

    def verify_packed(self, instances: bytes, proofs: bytes, n: int, mode: str = "merged", rng=None, weights=None,
                      pre_status: Optional[Sequence[int]] = None) -> List[int]:
        """The batch in wire form: `instances` = n x (vec_R | vec_S | vec_T | vec_U) encodings, `proofs` = n x
        crs.proof_bytes.  Returns the per-proof status (0 = valid, else a reject code of REJECT_NAMES)."""
        import time

        crs, ctx = self.crs, self.ctx
        L, C = crs.points_per_proof, crs.ncrs
        t1 = time.perf_counter()
        if weights is None:
            weights = self.draw_weights(n, rng)
        assert len(instances) == n * 4 * crs.ell * 48 and len(proofs) == n * crs.proof_bytes and len(weights) == n * N_WEIGHTS * 32
        t2 = t2b = time.perf_counter()
        bounds = [(lo, min(lo + self.chunk, n)) for lo in range(0, n, self.chunk)]
        if len(bounds) <= 1:
            pstat, decoded = self.decompress_on_gpu(instances, proofs, n)
            t2b = time.perf_counter()
            prep = self.prepare(instances, proofs, n, weights=weights, decoded=decoded, staging=self._device_buffers(n)["host"])
        else:
            # two-stage pipeline over sub-batches: a worker thread drives the GPU (gather, H2D, decompress, D2H) while
            # this thread runs the native front-end (all cores) on the sub-batches already decoded
            import queue
            import threading

            host = self._device_buffers(n)["host"]
            pstat, decoded = host["pstat"].buf, host["decoded"].buf
            done: "queue.Queue" = queue.Queue()

            def gpu_stage():
                try:
                    for lo, hi in bounds:
                        self.decompress_on_gpu(instances, proofs, n, lo, hi)
                        done.put((lo, hi))
                except BaseException as e:                      # surfaced in the consumer
                    done.put(e)
                finally:
                    done.put(None)

            if self._gpu_thread is None:                       # persistent: a thread's first HIP call is expensive
                self._gpu_jobs = queue.Queue()

                def loop(jobs=self._gpu_jobs):
                    while True:
                        job = jobs.get()
                        if job is None:
                            return
                        job()

                self._gpu_thread = threading.Thread(target=loop, daemon=True)
                self._gpu_thread.start()
            self._gpu_jobs.put(gpu_stage)
            prep = Prepared(crs, n, False, host)
            finished = False
            try:
                for _ in bounds:
                    r = done.get()
                    if r is None:
                        finished = True
                        raise N.NativeError("GPU stage ended early")
                    if isinstance(r, BaseException):
                        raise r
                    lo, hi = r
                    rc = N.cg1_shuffle_prepare(crs.handle, hi - lo, _addr(instances) + lo * 4 * crs.ell * 48,
                                               _addr(proofs) + lo * crs.proof_bytes, _addr(weights) + lo * N_WEIGHTS * 32,
                                               host["decoded"].ptr + lo * 768, 768, host["wire"].ptr + lo * L * 48,
                                               host["sc"].ptr + lo * L * 32, ctypes.addressof(prep.crs_scalars32) + lo * C * 32,
                                               ctypes.addressof(prep.status) + lo * 4, None, self.threads)
                    if rc:
                        raise N.NativeError(f"cg1_shuffle_prepare failed ({rc})")
            finally:
                while not finished and done.get() is not None:     # wait for the GPU stage before touching its buffers
                    pass
        ctx.check(N.cg1_shuffle_apply_point_status(prep.status, pstat, n, L, prep.scalars32, prep.crs_scalars32, C))
        for i, s in enumerate(pre_status or ()):
            if s:
                prep.status[i] = s
                ctypes.memset(ctypes.addressof(prep.scalars32) + i * L * 32, 0, L * 32)
                ctypes.memset(ctypes.addressof(prep.crs_scalars32) + i * C * 32, 0, C * 32)
        t3 = time.perf_counter()
        # (pipelined batches: gpu_decompress_s is hidden inside prepare_s)
        self.last_stats = {"weights_s": t2 - t1, "gpu_decompress_s": t2b - t2, "prepare_s": t3 - t2b, "pipelined": len(bounds) > 1}
        status = self.check_prepared(prep, mode, points_on_device=True)
        self.last_status = status
        self.last_stats["total_s"] = time.perf_counter() - t1
        return status


def is_valid_whisk_shuffle_proof(crs, pre_shuffle_trackers, post_shuffle_trackers, whisk_shuffle_proof_bytes, ctx=None) -> bool:
    """Drop-in for IsValidWhiskShuffleProof (whisk_interface.py:72-87) -- a batch of one."""
    v = crs if isinstance(crs, ShuffleBatchVerifier) else ShuffleBatchVerifier(crs, ctx)
    return v.verify_many([(pre_shuffle_trackers, post_shuffle_trackers, whisk_shuffle_proof_bytes)])[0]
