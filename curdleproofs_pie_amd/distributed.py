"""Multi-GPU MSM: one process per GPU, one tiny exchange step.  No PyTorch: the exchange lives behind the
C ABI (cg1_comm_*, csrc/comm.cpp) -- an RCCL all-gather over xGMI on a GPU node, a TCP star on the loopback interface when
several ranks rehearse on one GPU or on a CPU-only box.

A single large MSM is partitioned across the ranks of one node either
  * by WINDOW  ("windows", what BASELINE.json's north_star names): rank g owns the signed-digit windows
    w = g (mod world); every rank holds all n points/scalars and returns  sum_{w in g} 2^(c w) S_w ; or
  * by POINT   ("points"): rank g owns points [g*n/world, (g+1)*n/world) with all windows; or
  * by both    ("hybrid", `shard_layout`).
Either way the partials are ONE G1 point per rank.  RCCL has no elliptic-curve reduction operator, so the
"all-reduce of partial G1 sums" is an all-gather of `world` 144-byte blobs (latency-bound: ~1 KB) followed by world-1 host
G1 additions on every rank -- every rank ends with the same, bit-exact result (G1 addition is commutative/associative and
the final encoding is canonical).

Rendezvous (one node): rank 0 listens on an ephemeral loopback port and publishes "port nonce" in a small file; the other
ranks poll the file and connect.  The file's path comes from CG1_RDZV_FILE (bench.py's own launcher sets it) or is derived
from what every rank of one launch shares: the launcher's pid (all ranks are its children) and MASTER_PORT (torchrun sets it).
"""
from __future__ import annotations

import os
import secrets
import struct
import tempfile
import time
from typing import List, Optional

from . import _native as N

Comm = N.Comm


def sum_blobs(blobs: List[bytes]) -> bytes:
    """Host G1 sum of point blobs (the reduction operator of the 'all-reduce')."""
    import ctypes

    acc = ctypes.create_string_buffer(N.POINT_BYTES)
    N.cg1_identity(acc)
    for b in blobs:
        N.cg1_add(acc, acc.raw, bytes(b))
    return acc.raw


def _private_dir() -> str:
    """A directory only this user can enter: the rendezvous file holds the hub's port AND the only authenticator (the nonce), so it
    must not be readable -- and its name not pre-creatable -- by other local users."""
    import stat

    uid = os.getuid()
    run = os.environ.get("XDG_RUNTIME_DIR")
    cands = ([os.path.join(run, "cg1_rdzv")] if run else []) + [os.path.join(tempfile.gettempdir(), "cg1_rdzv_%d" % uid)]
    err = None
    for d in cands:
        try:
            os.makedirs(d, mode=0o700, exist_ok=True)
            st = os.lstat(d)
            if stat.S_ISDIR(st.st_mode) and st.st_uid == uid and not (st.st_mode & 0o077):
                return d
            err = "%s is not a private directory of uid %d" % (d, uid)
        except OSError as e:
            err = str(e)
    raise N.NativeError("no private directory for the rendezvous file (%s); set CG1_RDZV_FILE" % err)


def default_rendezvous_file() -> str:
    explicit = os.environ.get("CG1_RDZV_FILE")
    if explicit:
        return explicit
    return os.path.join(_private_dir(), "rdzv_%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0")))


def _read_rendezvous(path: str):
    """(port, nonce) from a rendezvous file this user wrote (regular file, own uid, no group / other access, no symlink), else None."""
    import stat

    try:
        fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
    except OSError:
        return None
    try:
        st = os.fstat(fd)
        if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            return None
        port, nonce = (int(x) for x in os.read(fd, 128).split())
        return port, nonce
    except (OSError, ValueError):
        return None
    finally:
        os.close(fd)


def init_comm(rank: int, world: int, rendezvous_file: Optional[str] = None, timeout_s: float = 600.0) -> "N.Comm":
    """Connect the control channel of this rank (collective: every rank of the job calls it)."""
    comm = N.Comm(rank, world)
    if world == 1:
        return comm
    path = rendezvous_file or default_rendezvous_file()
    deadline = time.monotonic() + timeout_s
    if rank == 0:
        nonce = secrets.randbits(63)
        tmp = "%s.%d.%x.tmp" % (path, os.getpid(), secrets.randbits(32))
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)      # never through a planted name or symlink
        try:
            os.write(fd, b"%d %d\n" % (comm.port, nonce))
        finally:
            os.close(fd)
        os.replace(tmp, path)                       # atomic: a reader sees the old file, no file, or the whole new one
        try:
            comm.check(comm.connect("", 0, nonce, int(timeout_s * 1000)))
        finally:
            try:
                os.unlink(path)
            except OSError:
                pass
        return comm
    while True:
        got = _read_rendezvous(path)
        port, nonce = got if got else (None, None)
        if port:
            left_ms = max(1000, int((deadline - time.monotonic()) * 1000))
            rc = comm.connect("127.0.0.1", port, nonce, left_ms)
            if rc == N.OK:
                return comm
        if time.monotonic() > deadline:
            comm.check(N.ERR_COMM if not port else rc)
            raise N.NativeError("rendezvous timed out: %s never appeared" % path)
        time.sleep(0.05)                            # no file yet, a stale file of an earlier launch, or not the hub: look again


def init_from_env(ctx: Optional["N.Context"] = None, transport: str = "auto", timeout_s: float = 600.0) -> "N.Comm":
    """RANK / WORLD_SIZE (as torch.distributed.run and bench.py's own launcher export them) -> a connected communicator.
    transport: "rccl" attaches RCCL on ctx's device (needs one GPU per rank), "socket" keeps the TCP star, "auto" = rccl when a
    context is given."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    comm = init_comm(rank, world, timeout_s=timeout_s)
    if transport == "rccl" or (transport == "auto" and ctx is not None):
        if ctx is None:
            raise ValueError("the RCCL transport needs the rank's Context")
        comm.attach_rccl(ctx)
    return comm


def all_reduce_g1(partial_blob: bytes, comm: Optional["N.Comm"] = None) -> bytes:
    """All-gather every rank's partial G1 point and add them; returns the same blob on all ranks (one collective)."""
    if comm is None or comm.world == 1:
        return bytes(partial_blob)
    return comm.allreduce_g1(partial_blob)


def max_over_ranks(value: float, comm: Optional["N.Comm"]) -> List[float]:
    """Every rank's value (control channel); bench.py takes the max of the per-rank clocks."""
    if comm is None or comm.world == 1:
        return [float(value)]
    return [struct.unpack("<d", b)[0] for b in comm.allgather(struct.pack("<d", float(value)), host_only=True)]


def shard_layout(rank: int, world: int, mode: str = "hybrid", window_groups: int = 2):
    """-> (window_rank, window_groups W, point_rank, point_groups P) with W * P == world.

    "windows": W = world (every rank holds ALL points, owns the windows w = rank mod world) -- replicates the
               per-point work (k_prepare_points, digit extraction) world times: 4.25 ms per rank at world = 8;
    "points":  P = world (every rank holds 1/world of the points, all windows);
    "hybrid":  W = window_groups (default 2) window-bucket groups x P = world / W point groups: window buckets are still
               sharded across GPUs, but a rank only touches 1/P of the points (3.47 ms per rank at world = 8 on MI355X,
               tools/gpu_shard_emulation.py).  Falls back to "windows" when world is not a multiple of window_groups."""
    if mode == "windows" or (mode == "hybrid" and (world % window_groups or world < window_groups)):
        return rank, world, 0, 1
    if mode == "points":
        return 0, 1, rank, world
    if mode != "hybrid":
        raise ValueError(f"unknown shard mode {mode!r}")
    W = window_groups
    return rank % W, W, rank // W, world // W


def sharded_msm(ctx: "N.Context", d_points, d_scalars, n: int, rank: int, world: int, window_c: int = 16,
                mode: str = "windows", comm: Optional["N.Comm"] = None) -> bytes:
    """This rank's share of one MSM, then the G1 all-reduce.

    mode "windows": d_points/d_scalars hold ALL n terms on every rank.
    mode "points" : d_points/d_scalars hold only this rank's shard of n terms.
    mode "hybrid" : d_points/d_scalars hold the n terms of this rank's point group (see shard_layout).
    """
    if mode == "windows":
        part = ctx.msm_device(d_points, d_scalars, n, window_c=window_c, shard_rank=rank, shard_world=world)
    elif mode == "points":
        part = ctx.msm_device(d_points, d_scalars, n, window_c=window_c)
    elif mode == "hybrid":              # d_points/d_scalars hold this rank's POINT GROUP (n terms); windows split inside the group
        wr, W, _, _ = shard_layout(rank, world, "hybrid")
        part = ctx.msm_device(d_points, d_scalars, n, window_c=window_c, shard_rank=wr, shard_world=W)
    else:
        raise ValueError(f"unknown shard mode {mode!r}")
    return all_reduce_g1(part, comm)


def sharded_msm_batch(jobs_affine, rank: int, world: int, comm: Optional["N.Comm"] = None, compute=None) -> List[bytes]:
    """Many INDEPENDENT MSMs (e.g. one final accumulator MSM per proof; BASELINE config 5: 16384 proofs over
    8 GPUs) sharded job-per-GPU: rank g computes jobs g, g+world, ... with the regime-B batched kernels, then the
    per-job result blobs are all-gathered so every rank holds all results.  Embarrassingly parallel: the only
    collective is that final gather of 144 bytes per job (no data-path exchange).

    jobs_affine: list of (points_affine96: bytes, scalars32: bytes, n: int).
    compute: callable(list_of_jobs) -> list of blobs for this rank's share; defaults to the GPU batched path
             (curdleproofs_pie_amd has no CPU path; the parameter exists so the CPU tests can exercise the
             sharding/gather logic on a CPU-only box with the host operators).
    """
    mine = list(range(rank, len(jobs_affine), world))
    my_jobs = [jobs_affine[i] for i in mine]
    if compute is None:
        def compute(js):
            if not js:
                return []
            offs = [0]
            for _, _, n in js:
                offs.append(offs[-1] + n)
            return N.default_context().msm_batched_host(b"".join(p for p, _, _ in js), b"".join(s for _, s, _ in js), offs)
    my_blobs = compute(my_jobs)
    assert len(my_blobs) == len(my_jobs)
    out = [None] * len(jobs_affine)
    if world == 1 or comm is None:
        for i, b in zip(mine, my_blobs):
            out[i] = bytes(b)
        return out
    # fixed-size gather: rank g owns jobs g, g + world, ... (known to every rank, so no indices travel), each rank
    # sends ceil(jobs / world) slots of 144 bytes, unused slots zero
    width = (len(jobs_affine) + world - 1) // world
    for b in my_blobs:
        assert len(b) == N.POINT_BYTES
    local = b"".join(bytes(b) for b in my_blobs).ljust(width * N.POINT_BYTES, b"\0")
    for g, raw in enumerate(comm.allgather(local)):
        for k, i in enumerate(range(g, len(jobs_affine), world)):
            out[i] = raw[k * N.POINT_BYTES:(k + 1) * N.POINT_BYTES]
    return out


def sharded_verify(verifier, instances: bytes, proofs: bytes, n: int, rank: int, world: int, comm: Optional["N.Comm"] = None,
                   mode: str = "merged", verify=None) -> List[int]:
    """BASELINE config 5 (16 384 Whisk shuffle verifications over 8 GPUs): proof-per-GPU sharding.  Rank g verifies the
    contiguous slice [g*n/world, (g+1)*n/world) of the packed batch with its own `ShuffleBatchVerifier` (one merged MSM
    per rank); the per-proof status codes (0 = valid) are all-gathered so every rank returns the full list.
    No data-path collective: the only exchange is that gather of one small integer per proof (control channel).

    verify: callable(instances_slice, proofs_slice, m) -> list of m status ints; defaults to verifier.verify_packed
            (the parameter exists so the CPU tests can exercise the slicing / gather logic on a CPU-only box).
    """
    crs = verifier.crs
    inst_b, proof_b = 4 * crs.ell * 48, crs.proof_bytes
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    if verify is None:
        verify = lambda a, b, m: verifier.verify_packed(a, b, m, mode=mode)
    mine = list(verify(instances[lo * inst_b: hi * inst_b], proofs[lo * proof_b: hi * proof_b], hi - lo)) if hi > lo else []
    assert len(mine) == hi - lo
    if world == 1 or comm is None:
        return mine
    # fixed-size gather (n/world differs by at most one between ranks: pad to the maximum)
    width = (n + world - 1) // world
    local = struct.pack("<%di" % width, *(list(mine) + [-1] * (width - len(mine))))
    out: List[int] = []
    for g, raw in enumerate(comm.allgather(local, host_only=True)):
        cnt = (n * (g + 1)) // world - (n * g) // world
        out.extend(struct.unpack("<%di" % width, raw)[:cnt])
    return out
