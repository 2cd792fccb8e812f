"""The N>1 combine step on CPU, through the product's own exchange (cg1_comm_*, csrc/comm.cpp; no PyTorch): world_size-2 and -4
ranks connect their TCP control channel through a rendezvous file, each holds one partial G1 sum (a share of the terms of one
MSM, computed here with the host operators), all-gathers the 144-byte blobs and adds them -- exactly what
curdleproofs_pie_amd.distributed does with RCCL attached on the GPU box.  Every rank must end with the same group element as
the single-process oracle."""
import os
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rdzv():
    return os.path.join(tempfile.mkdtemp(prefix="cg1_rdzv_test_"), "rdzv")


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    import ctypes
    import random

    from curdleproofs_pie_amd import _native as N
    from curdleproofs_pie_amd.distributed import all_reduce_g1, init_comm
    from curdleproofs_pie_amd.py_arkworks_bls12381 import G1Point, Scalar

    comm = init_comm(rank, world, rendezvous_file=port, timeout_s=120)
    assert comm.world_seen == world and comm.transport == "socket"
    rng = random.Random(2024)          # same inputs on every rank
    n = 24
    bases = [G1Point() * Scalar(rng.randint(1, 2 ** 200)) for _ in range(n)]
    scalars = [rng.randint(0, 2 ** 255 - 20) for _ in range(n)]
    part = G1Point.identity()
    if mode == "hybrid":               # W window groups x P point groups (shard_layout): windows wr mod W of the point slice pr
        from curdleproofs_pie_amd.distributed import shard_layout
        wr, W, pr, P = shard_layout(rank, world, "hybrid")
        lo, hi = pr * n // P, (pr + 1) * n // P
        for b, s in zip(bases[lo:hi], scalars[lo:hi]):
            mine = sum(((s >> (16 * w)) & 0xFFFF) << (16 * w) for w in range(16) if w % W == wr)
            part = part + b * Scalar(mine)
    elif mode == "points":             # rank owns a contiguous slice of the terms
        lo, hi = rank * n // world, (rank + 1) * n // world
        for b, s in zip(bases[lo:hi], scalars[lo:hi]):
            part = part + b * Scalar(s)
    else:                              # rank owns the 16-bit windows w = rank (mod world) of every scalar
        for b, s in zip(bases, scalars):
            mine = sum(((s >> (16 * w)) & 0xFFFF) << (16 * w) for w in range(16) if w % world == rank)
            part = part + b * Scalar(mine)
    total = all_reduce_g1(part._b, comm)
    out = ctypes.create_string_buffer(48)
    N.cg1_compress(out, total)
    q.put((rank, out.raw.hex()))
    comm.barrier()
    comm.close()


@pytest.mark.parametrize("mode", ["windows", "points"])
def test_two_rank_g1_all_reduce(native_lib, mode):
    import random

    import multiprocessing as mp

    from oracle import bls12_381 as O

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _rdzv()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = random.Random(2024)
    n = 24
    ks = [rng.randint(1, 2 ** 200) for _ in range(n)]
    scalars = [rng.randint(0, 2 ** 255 - 20) for _ in range(n)]
    want = O.g1_compress(O.g1_mul(O.G1_GEN, sum(k * (s % O.R) for k, s in zip(ks, scalars)) % O.R)).hex()
    assert got[0] == got[1] == want


@pytest.mark.parametrize("world", [4, 8])
def test_hybrid_all_reduce_over_four_and_eight_ranks(native_lib, world):
    """2 window groups x (world / 2) point groups: every rank ends with the single-process result.  world = 8 is the driver's SCALE shape
    (a GPU rehearsal of it is not possible: the pool admits 6 processes per card), here over the TCP transport on CPU."""
    import random

    import multiprocessing as mp

    from curdleproofs_pie_amd.distributed import shard_layout
    from oracle import bls12_381 as O

    assert [shard_layout(r, 8, "hybrid") for r in range(8)] == [(r % 2, 2, r // 2, 4) for r in range(8)]
    assert shard_layout(3, 8, "windows") == (3, 8, 0, 1) and shard_layout(3, 8, "points") == (0, 1, 3, 8)
    assert shard_layout(2, 3, "hybrid") == (2, 3, 0, 1)                  # odd world: pure window sharding
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _rdzv()
    procs = [ctx.Process(target=_worker, args=(r, world, port, "hybrid", q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = random.Random(2024)
    n = 24
    ks = [rng.randint(1, 2 ** 200) for _ in range(n)]
    scalars = [rng.randint(0, 2 ** 255 - 20) for _ in range(n)]
    want = O.g1_compress(O.g1_mul(O.G1_GEN, sum(k * (s % O.R) for k, s in zip(ks, scalars)) % O.R)).hex()
    assert len(got) == world and set(got.values()) == {want}


def _worker_batch(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import ctypes
    import random

    from curdleproofs_pie_amd import _native as N
    from curdleproofs_pie_amd.distributed import init_comm, sharded_msm_batch
    from curdleproofs_pie_amd.py_arkworks_bls12381 import G1Point, Scalar, points_to_affine96

    comm = init_comm(rank, world, rendezvous_file=port, timeout_s=120)
    assert comm.world_seen == world and comm.transport == "socket"
    rng = random.Random(77)
    jobs, truth = [], []
    for n in (3, 1, 0, 5, 2):          # 5 independent "proofs"
        ks = [rng.randint(1, 2 ** 100) for _ in range(n)]
        ss = [rng.randint(0, 2 ** 250) for _ in range(n)]
        pts = [G1Point() * Scalar(k) for k in ks]
        jobs.append((points_to_affine96(pts), b"".join(s.to_bytes(32, "little") for s in ss), n))
        truth.append(sum(k * s for k, s in zip(ks, ss)))

    def host_compute(js):              # CPU stand-in for the GPU batched kernels (test only)
        out = []
        for p96, s32, n in js:
            acc = G1Point.identity()
            for i in range(n):
                b = ctypes.create_string_buffer(N.POINT_BYTES)
                assert N.cg1_from_affine96(b, p96[96 * i: 96 * i + 96], 1) == 0
                acc = acc + G1Point._from_blob(b.raw) * Scalar(int.from_bytes(s32[32 * i: 32 * i + 32], "little"))
            out.append(acc._b)
        return out

    blobs = sharded_msm_batch(jobs, rank, world, comm=comm, compute=host_compute)
    res = []
    for b in blobs:
        o = ctypes.create_string_buffer(48)
        N.cg1_compress(o, b)
        res.append(o.raw.hex())
    q.put((rank, res, truth))
    comm.barrier()
    comm.close()


def test_two_rank_job_sharding(native_lib):
    import multiprocessing as mp

    from oracle import bls12_381 as O

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _rdzv()
    procs = [ctx.Process(target=_worker_batch, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, r0, truth), (_, r1, _) = got
    assert r0 == r1
    assert r0 == [O.g1_compress(O.g1_mul(O.G1_GEN, t % O.R)).hex() for t in truth]


def _worker_verify(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import json
    import random

    from curdleproofs_pie_amd.distributed import init_comm, sharded_verify
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
    from oracle.shuffle_check import oracle_verdicts

    comm = init_comm(rank, world, rendezvous_file=port, timeout_s=120)
    assert comm.world_seen == world and comm.transport == "socket"
    with open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")) as f:
        case = json.load(f)["cases"][1]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_shuffle_verifier import apply_edits

    variants = case["variants"][:3] + case["variants"][-4:]            # 7 proofs over 2 ranks: uneven slices (3 + 4)
    v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), threads=2)
    inst, proofs, _ = v.pack([apply_edits(case, x["edits"]) for x in variants])

    def host_verify(a, b, m):          # CPU stand-in for the GPU half (test only): front-end + CPU-oracle MSM
        prep = v.prepare(a, b, m, rng=random.Random(5 + rank))
        return [0 if ok else 6 for ok in oracle_verdicts(v, prep)]

    status = sharded_verify(v, inst, proofs, len(variants), rank, world, comm=comm, verify=host_verify)
    q.put((rank, status, [x["accepts"] for x in variants]))
    comm.barrier()
    comm.close()


def test_two_rank_proof_sharding(native_lib):
    """BASELINE config 5's structure: proofs sharded per rank, verdicts all-gathered (socket transport; CPU stand-in for the GPU half)."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _rdzv()
    procs = [ctx.Process(target=_worker_verify, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, s0, want), (_, s1, _) = got
    assert s0 == s1 and len(s0) == len(want)
    assert [s == 0 for s in s0] == want


def test_rendezvous_survives_a_stale_file(native_lib):
    """A rendezvous file left behind by an earlier launch (dead port, other nonce) must not wedge the ranks."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    path = _rdzv()
    with open(path, "w") as f:
        f.write("1 12345\n")                      # nobody listens on port 1
    procs = [ctx.Process(target=_worker, args=(r, 2, path, "points", q)) for r in (1, 0)]
    procs[0].start()                              # rank 1 first: it meets the stale file
    import time
    time.sleep(1.0)
    procs[1].start()
    got = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1]


def test_collective_mismatch_is_an_error_not_a_hang(native_lib):
    """Ranks that disagree on the payload size of a collective get CG1_ERR_COMM, not a deadlock."""
    import threading

    from curdleproofs_pie_amd import _native as N
    from curdleproofs_pie_amd.distributed import init_comm

    path = _rdzv()
    res = {}

    def run(rank, nbytes):
        c = init_comm(rank, 2, rendezvous_file=path, timeout_s=60)
        c.set_timeout(5000)
        try:
            c.allgather(b"x" * nbytes, host_only=True)
            res[rank] = "ok"
        except N.NativeError as e:
            res[rank] = str(e)
        c.close()

    ts = [threading.Thread(target=run, args=(r, 4 + r)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert "mismatch" in res[0] and res[1] != "ok"
