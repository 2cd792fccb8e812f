"""BASELINE config 3 at its own size: a batch of 1024 Whisk shuffle verifications (ell = 124 + 4 blinders) made of the 1024
DISTINCT proofs of tests/golden/shuffle_batch_ell124{,_more}.bin (reference prover, tests/golden/gen_shuffle_batch.py) with
tampered proofs at known slots; verdicts must equal the ones the reference's IsValidWhiskShuffleProof
(whisk_interface.py:72-87 -> curdleproofs.py:162-248) returned when the fixture was made.

CPU: the native front-end's statements evaluated by the CPU oracle (all 12 tampered variants + 4 valid proofs).
GPU: the product path -- verify_packed in both modes and verify_stream."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from batch_fixture import ShuffleBatch  # noqa: E402


@pytest.fixture(scope="module")
def fx():
    f = ShuffleBatch()
    assert f.count == 1024 and f.ell == 124 and len(set(f.proofs)) == len(set(f.instances)) == f.count      # all distinct
    assert [t["accepts"] for t in f.tampered].count(False) >= 10
    return f


def test_front_end_statements_match_reference_verdicts_cpu(native_lib, fx):
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
    from oracle.shuffle_check import oracle_verdicts

    v = ShuffleBatchVerifier(fx.crs)
    slots = {2 * i + 1: i for i in range(len(fx.tampered))}            # every tampered variant, valid proofs in between
    n = 2 * len(fx.tampered) + 2
    inst, proofs, want = fx.tiled(n, slots)
    prep = v.prepare(inst, proofs, n, rng=random.Random(3))
    assert oracle_verdicts(v, prep) == want


SLOTS_1024 = {0: 0, 7: 3, 64: 1, 65: 4, 500: 2, 511: 6, 512: 7, 777: 8, 800: 9, 1000: 10, 1023: 5, 300: 11}


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["merged", "independent"])
def test_batch_1024_verdicts_equal_reference(native_lib, fx, mode):
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    v = ShuffleBatchVerifier(fx.crs, native_lib.Context(0))
    inst, proofs, want = fx.tiled(1024, SLOTS_1024)
    assert want.count(False) == 11
    status = v.verify_packed(inst, proofs, 1024, mode=mode, rng=random.Random(5))
    assert [s == 0 for s in status] == want
    if mode == "merged":
        assert v.last_stats["merged_ok"] is False                     # the merged check failed, culprits named by the fallback
        clean, cproofs, cwant = fx.tiled(1024)
        assert v.verify_packed(clean, cproofs, 1024, rng=random.Random(6)) == [0] * 1024
        assert v.last_stats["merged_ok"] is True and v.last_stats["points"] == 1024 * v.crs.points_per_proof + v.crs.ncrs
    v.close()


@pytest.mark.gpu
def test_stream_of_1024_batches(native_lib, fx):
    """verify_stream: three consecutive 1024-proof batches (clean, tampered, clean) overlapped across the pipeline stages,
    fresh OS-random weights (no rng): verdicts per batch equal the fixture's."""
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    v = ShuffleBatchVerifier(fx.crs, native_lib.Context(0))
    clean = fx.tiled(1024)
    bad = fx.tiled(1024, SLOTS_1024)
    got = list(v.verify_stream([(clean[0], clean[1], 1024), (bad[0], bad[1], 1024), (clean[0], clean[1], 1024)]))
    assert [[s == 0 for s in st] for st in got] == [clean[2], bad[2], clean[2]]
    v.close()


@pytest.mark.gpu
def test_config5_16384_proofs_sharded_proof_per_gpu(native_lib, fx):
    """BASELINE config 5 at its size: 16 384 ell = 124 verifications sharded proof-per-GPU over 8 ranks
    (whisk_interface.py:72-87 per proof; distributed.sharded_verify per rank) -- the 8 rank slices of 2048 proofs run one after
    another on this GPU, each through its own streamed verifier calls, tampered proofs in EVERY slice (at slice-dependent slots);
    the gathered verdicts must equal the fixture's (the reference verifier's).  Every distinct proof of the fixture is used."""
    from curdleproofs_pie_amd.distributed import sharded_verify
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    n, world = 16384, 8
    slots = {}
    for g in range(world):                                              # 5 tampered proofs per slice, the variants rotating
        for j in range(5):
            slots[g * (n // world) + (131 * j + 17 * g) % (n // world)] = (5 * g + j) % len(fx.tampered)
    inst, proofs, want = fx.tiled(n, slots)
    assert want.count(False) >= 30 and all(want[g * 2048: (g + 1) * 2048].count(False) >= 3 for g in range(world))
    v = ShuffleBatchVerifier(fx.crs, native_lib.Context(0))

    def verify(a, b, m):                                                # one rank's slice as a stream of 1024-proof batches
        ib, pb = fx.inst_bytes, fx.proof_bytes
        batches = [(a[lo * ib: (lo + 1024) * ib], b[lo * pb: (lo + 1024) * pb], min(1024, m - lo)) for lo in range(0, m, 1024)]
        out = []
        for st in v.verify_stream(batches):
            out.extend(st)
        return out

    # eight ranks as eight threads of this process, each with its own communicator (the library's TCP control channel, world 8);
    # the GPU is taken in turns (one verifier), the slicing and the verdict gather are sharded_verify's own
    import tempfile
    import threading

    from curdleproofs_pie_amd.distributed import init_comm

    path = os.path.join(tempfile.mkdtemp(prefix="cg1_cfg5_"), "rdzv")
    turn = threading.Lock()

    def locked_verify(a, b, m):
        with turn:
            return verify(a, b, m)

    results, errors = {}, []

    def rank_main(rank):
        try:
            comm = init_comm(rank, world, rendezvous_file=path, timeout_s=300)
            comm.set_timeout(900000)
            results[rank] = sharded_verify(v, inst, proofs, n, rank, world, comm=comm, verify=locked_verify)
            comm.barrier()
            comm.close()
        except BaseException as e:
            errors.append((rank, repr(e)))

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=1500)
    assert not errors, errors
    for rank in range(world):
        assert [s == 0 for s in results[rank]] == want, rank
    v.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pipelines", [3, 1])
def test_device_front_end_stream_verdicts_equal_reference(native_lib, fx, pipelines):
    """The whole verifier with its front-end on the GPU (device_front_end=True: transcript, D / A', challenge algebra in
    k_shuffle_front_end): a stream of 1024-proof batches -- clean, tampered at known slots, clean, tampered -- with several batches in
    flight, over one pipeline and over two that take the batches in turn; verdicts equal the fixture's (the reference verifier's), in
    order, in both MSM modes; a consumer that stops early leaves nothing running."""
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    v = ShuffleBatchVerifier(fx.crs, native_lib.Context(0), device_front_end=True, pipelines=pipelines)
    assert v.pipelines == pipelines
    clean = fx.tiled(1024)
    bad = fx.tiled(1024, SLOTS_1024)
    seq = [clean, bad, clean, bad, clean, clean, bad, clean, clean]
    got = list(v.verify_stream([(x[0], x[1], 1024) for x in seq]))
    assert [[s == 0 for s in st] for st in got] == [x[2] for x in seq]
    st = v.verify_packed(bad[0], bad[1], 1024, mode="independent", rng=random.Random(9))
    assert [s == 0 for s in st] == bad[2]
    small = fx.tiled(37, {5: 2, 36: 7})
    assert [s == 0 for s in v.verify_packed(small[0], small[1], 37)] == small[2]
    gen = v.verify_stream([(x[0], x[1], 1024) for x in seq])
    assert [s == 0 for s in next(gen)] == clean[2]
    gen.close()                                               # abandoned after the first verdict
    assert [[s == 0 for s in st] for st in v.verify_stream([(bad[0], bad[1], 1024)])] == [bad[2]]
    v.close()


@pytest.mark.gpu
@pytest.mark.parametrize("device_front_end", [False, True])
def test_large_call_is_streamed_in_pieces(native_lib, fx, device_front_end):
    """verify_packed with more than 2 * SPLIT proofs (4100 here; BASELINE config 5 hands a node 16 384) goes through the stream in
    pieces of SPLIT: same verdicts, at the tampered slots of every piece, with caller-supplied weights and pre-rejected slots."""
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    n = 2 * ShuffleBatchVerifier.SPLIT + 4
    slots = {3: 0, 2047: 5, 2048: 7, 4095: 2, 4096: 9, n - 1: 11}
    inst, proofs, want = fx.tiled(n, slots)
    v = ShuffleBatchVerifier(fx.crs, native_lib.Context(0), device_front_end=device_front_end)
    got = v.verify_packed(inst, proofs, n)
    assert [s == 0 for s in got] == want and len(v.last_status) == n
    w = v.draw_weights(n, random.Random(3))
    pre = [0] * n
    pre[100] = pre[3000] = 1                                  # REJECT_LENGTH from pack()
    got = v.verify_packed(inst, proofs, n, weights=w, pre_status=pre)
    want2 = list(want)
    want2[100] = want2[3000] = False
    assert [s == 0 for s in got] == want2 and got[100] == 1 and got[3000] == 1
    v.close()


@pytest.mark.gpu
def test_stream_travels_in_coalesced_batches_when_queues_are_scarce(native_lib, fx):
    """With few hardware queues (the runtime's default 4) consecutive batches of a stream share one internal batch -- one front-end launch,
    one decoding pass, one merged MSM -- and every caller's batch still gets its own verdicts, in order: seven batches of 300 / 512
    proofs, tampered proofs at known slots of three of them, coalesced up to 1 100 proofs; an oversized batch and one with its own
    weights pass through on their own."""
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    v = ShuffleBatchVerifier(fx.crs, native_lib.Context(0), device_front_end=True, pipelines=1, fe_lanes=1, coalesce=1100)
    plain = ShuffleBatchVerifier(fx.crs, native_lib.Context(0), device_front_end=True, pipelines=1, fe_lanes=1, coalesce=0)
    sizes = [300, 512, 300, 300, 1200, 512, 300]
    slots = [{}, {5: 1, 511: 3}, {0: 2}, {}, {7: 4, 1100: 5}, {}, {299: 6}]
    batches, want = [], []
    for n, sl in zip(sizes, slots):
        inst, proofs, w = fx.tiled(n, sl)
        batches.append((inst, proofs, n))
        want.append(w)
    w5 = plain.draw_weights(512, random.Random(3))
    batches[5] = (batches[5][0], batches[5][1], 512, None, w5)       # brings its own weights: not merged with its neighbours
    got = list(v.verify_stream(iter(batches)))
    assert [len(st) for st in got] == sizes
    assert [[s == 0 for s in st] for st in got] == want
    assert [[s == 0 for s in st] for st in plain.verify_stream(iter(batches))] == want
    v.close(); plain.close()


@pytest.mark.gpu
def test_coalesced_stream_over_pipelines(native_lib, fx):
    """The same with several pipelines behind the coalescing verifier (24 hardware queues: the default shape), for limits that merge
    two caller batches, none at all (every internal batch = one caller batch: the children must not wait for a second one), and
    uneven sizes; the stream ends cleanly each time."""
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    sizes = [512, 512, 300, 700, 512, 512, 512, 100, 512]
    slots = [{}, {3: 1}, {}, {650: 2}, {}, {}, {511: 3}, {}, {0: 4}]
    batches, want = [], []
    for n, sl in zip(sizes, slots):
        inst, proofs, w = fx.tiled(n, sl)
        batches.append((inst, proofs, n))
        want.append(w)
    for limit in (1024, 600, 4096):
        v = ShuffleBatchVerifier(fx.crs, native_lib.Context(0), device_front_end=True, pipelines=2, fe_lanes=1, coalesce=limit)
        got = list(v.verify_stream(iter(batches)))
        assert [[s == 0 for s in st] for st in got] == want, limit
        v.close()
