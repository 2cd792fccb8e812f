"""GPU tests of the batch shuffle verifier: verdicts on the golden proofs (tests/golden/shuffle_vectors.json, verdicts
recorded from the reference's IsValidWhiskShuffleProof) through the C ABI -- GPU decompression + merged / independent MSMs."""
import json
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def ctx():
    from curdleproofs_pie_amd import _native as N

    return N.default_context()


def _items(case, variants):
    from test_shuffle_verifier import apply_edits

    return [apply_edits(case, x["edits"]) for x in variants]


@pytest.mark.parametrize("device_rows", [True, False], ids=["rows_on_device", "rows_on_host"])
@pytest.mark.parametrize("mode", ["merged", "independent"])
def test_golden_verdicts(gold, ctx, mode, device_rows):
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    for case in gold["cases"]:
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), ctx, device_rows=device_rows)
        got = v.verify_many(_items(case, case["variants"]), mode=mode, rng=random.Random(case["seed"]))
        want = [x["accepts"] for x in case["variants"]]
        assert got == want, [(x["name"], g, w, s) for x, g, w, s in zip(case["variants"], got, want, v.last_status) if g != w]


def test_all_valid_batch_takes_the_merged_path(gold, ctx):
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    c5, c6 = gold["cases"][4], gold["cases"][5]
    assert c5["ell"] == c6["ell"] == 124 and c5["crs"] != c6["crs"]
    for case in (c5, c6):
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), ctx)
        item = _items(case, [{"edits": []}])[0]
        got = v.verify_many([item] * 64, rng=random.Random(1))
        assert got == [True] * 64 and v.last_stats["merged_ok"] is True
        # one bad proof in the batch: merged check fails, the fallback names exactly that one
        bad = _items(case, [x for x in case["variants"] if x["name"] == "post_r[1] := other point"])[0]
        batch = [item] * 20 + [bad] + [item] * 11
        got = v.verify_many(batch, rng=random.Random(2))
        assert v.last_stats["merged_ok"] is False
        assert got == [True] * 20 + [False] + [True] * 11 and v.last_status[20] == 6
        # a proof rejected before any group arithmetic (non-canonical Fr) does not spoil the merged check of the others
        early = _items(case, [x for x in case["variants"] if x["name"] == "proof.x_final := r (non-canonical)"])[0]
        got = v.verify_many([item] * 5 + [early] + [item] * 2, rng=random.Random(3))
        assert got == [True] * 5 + [False] + [True] * 2 and v.last_stats["merged_ok"] is True and v.last_status[5] == 1


def test_pipelined_sub_batches_agree(gold, ctx):
    """chunk < batch: GPU decompression of sub-batch k+1 overlaps the host front-end of sub-batch k; same verdicts."""
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    case = gold["cases"][3]
    items = _items(case, case["variants"]) * 3
    want = [x["accepts"] for x in case["variants"]] * 3
    for chunk in (5, 16, 1000):
        v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), ctx, chunk=chunk)
        assert v.verify_many(items, rng=random.Random(chunk)) == want
        assert v.last_stats["pipelined"] == (chunk < len(items))


def test_stream_of_batches(gold, ctx):
    """verify_stream overlaps decompression / front-end / MSM of consecutive batches (three rotating buffer slots);
    batches of different sizes, some with invalid proofs: each batch gets exactly its own verdicts, in order."""
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    case = gold["cases"][2]
    v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), ctx, chunk=4)
    good = _items(case, [{"edits": []}])[0]
    variants = case["variants"]
    batches, want = [], []
    for b in range(7):
        items = [good] * (3 + b) + _items(case, variants[b: b + 2]) + [good] * (b % 3)
        inst, proofs, pre = v.pack(items)
        batches.append((inst, proofs, len(items), pre))
        want.append([True] * (3 + b) + [x["accepts"] for x in variants[b: b + 2]] + [True] * (b % 3))
    got = [[s == 0 for s in st] for st in v.verify_stream(batches, rng=random.Random(8))]
    assert got == want
    assert list(v.verify_stream([])) == []
    # a consumer that stops early must not strand the buffer slots of the batches still in flight
    for _ in range(4):
        gen = v.verify_stream(batches, rng=random.Random(9))
        assert [s == 0 for s in next(gen)] == want[0]
        gen.close()
    assert [[s == 0 for s in st] for st in v.verify_stream(batches[:4], rng=random.Random(10))] == want[:4]


def test_cross_crs_proof_is_rejected(gold, ctx):
    """A valid proof checked against another CRS of the same size must fail (every CRS slot matters)."""
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    c5, c6 = gold["cases"][4], gold["cases"][5]
    v = ShuffleBatchVerifier(bytes.fromhex(c6["crs"]), ctx)
    assert v.verify_many(_items(c5, [{"edits": []}]), rng=random.Random(3)) == [False]


def test_single_proof_wrapper_and_bad_shapes(gold, ctx):
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier, is_valid_whisk_shuffle_proof

    case = gold["cases"][1]
    pre, post, proof = _items(case, [{"edits": []}])[0]
    assert is_valid_whisk_shuffle_proof(bytes.fromhex(case["crs"]), pre, post, proof, ctx) is True
    assert is_valid_whisk_shuffle_proof(bytes.fromhex(case["crs"]), post, pre, proof, ctx) is False
    v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), ctx)
    assert v.verify_many([(pre, post, proof + b"xx"), (pre, post, proof[:-1]), (pre[:-1], post, proof), (pre, post, proof)]) == [True, False, False, True]
    assert v.verify_many([]) == []


def _rank_sharded_verify(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from curdleproofs_pie_amd import _native as N
    from curdleproofs_pie_amd.distributed import init_comm, sharded_verify
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
    from test_shuffle_verifier import apply_edits

    comm = init_comm(rank, world, rendezvous_file=port, timeout_s=300)      # the library's own exchange (socket transport: both ranks on one GPU)
    with open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")) as f:
        case = json.load(f)["cases"][2]
    v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), N.Context(0), threads=4)
    variants = case["variants"]
    inst, proofs, _ = v.pack([apply_edits(case, x["edits"]) for x in variants])
    status = sharded_verify(v, inst, proofs, len(variants), rank, world, comm=comm)
    q.put((rank, status, [x["accepts"] for x in variants]))
    comm.barrier()
    comm.close()


def test_two_ranks_share_a_batch_on_the_gpu(gold):
    """BASELINE config 5's structure with the real GPU path: two ranks (both on this GPU, cg1_comm_* socket transport) verify
    disjoint slices of one batch and all-gather the verdicts; every rank ends with the reference's verdict for every proof."""
    import multiprocessing as mp
    import tempfile

    port = os.path.join(tempfile.mkdtemp(prefix="cg1_rdzv_test_"), "rdzv")
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_rank_sharded_verify, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, s0, want), (_, s1, _) = got
    assert s0 == s1 and [s == 0 for s in s0] == want
