"""The same sharding / gather logic over a `gloo` process group (world_size 2, CPU): the functions of
curdleproofs_pie_amd.distributed only need an object with `.world`, `.allgather(bytes)` and `.allreduce_g1(blob)`, so a
test-side adapter over torch.distributed stands in for the product's communicator (cg1_comm_*, which is what runs in
production and in tests/test_distributed_socket.py -- the product itself never imports torch)."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class GlooComm:
    """torch.distributed (gloo) behind the interface of curdleproofs_pie_amd._native.Comm -- test infrastructure only."""

    def __init__(self, rank, world):
        import torch.distributed as dist

        dist.init_process_group("gloo", rank=rank, world_size=world)
        self.rank, self.world, self.transport = rank, world, "gloo"

    def allgather(self, data, host_only=False):
        import torch
        import torch.distributed as dist

        mine = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        out = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(out, mine)
        return [t.numpy().tobytes() for t in out]

    def allreduce_g1(self, blob):
        from curdleproofs_pie_amd.distributed import sum_blobs

        return sum_blobs(self.allgather(blob))

    def barrier(self):
        import torch.distributed as dist

        dist.barrier()

    def close(self):
        import torch.distributed as dist

        dist.destroy_process_group()


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import ctypes
    import random

    from curdleproofs_pie_amd import _native as N
    from curdleproofs_pie_amd.distributed import all_reduce_g1
    from curdleproofs_pie_amd.py_arkworks_bls12381 import G1Point, Scalar

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_distributed_gloo import GlooComm

    comm = GlooComm(rank, world)
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

    import torch.multiprocessing as mp

    from oracle import bls12_381 as O

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
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


def _worker_verify(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import json
    import random

    from curdleproofs_pie_amd.distributed import sharded_verify
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
    from oracle.shuffle_check import oracle_verdicts

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_distributed_gloo import GlooComm

    comm = GlooComm(rank, world)
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
    """BASELINE config 5's structure: proofs sharded per rank, verdicts all-gathered (gloo; CPU stand-in for the GPU half)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
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
