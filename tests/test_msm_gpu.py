"""Parity of the HIP MSM path (through the C ABI) against the CPU oracle.  Needs an MI355X.

Bar: bit-exact -- an MSM result is a unique group element with a unique canonical 48-byte compression
(what `G1Point.to_compressed_bytes()` returns in the reference, util.py:27-28)."""
import ctypes
import random

import pytest

from conftest import raw96
from oracle import bls12_381 as O
from oracle import c_oracle as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(native_lib):
    c = native_lib.Context(0)
    yield c
    c.close()


def compress_blob(N, blob):
    out = ctypes.create_string_buffer(48)
    N.cg1_compress(out, blob)
    return out.raw


def gpu_msm(N, ctx, pts96, sc32, n, **kw):
    dp, ds = ctx.alloc(max(96 * n, 96)), ctx.alloc(max(32 * n, 32))
    if n:
        dp.upload(pts96); ds.upload(sc32)
    try:
        return compress_blob(N, ctx.msm_device(dp, ds, n, **kw))
    finally:
        dp.free(); ds.free()


def test_golden_vectors(native_lib, ctx, golden):
    for case in golden:
        pts = [O.g1_decompress(bytes.fromhex(h)) for h in case["points"]]
        n = len(pts)
        p96 = b"".join(raw96(p) for p in pts)
        s32 = b"".join(bytes.fromhex(h) for h in case["scalars"])
        for c in (0, 5, 16, -13, -7):         # automatic plan, uniform widths, balanced plans (cmax = 13 / 7)
            assert gpu_msm(native_lib, ctx, p96, s32, n, window_c=c).hex() == case["expected"], (case["name"], c)
        # host-pointer entry point (what compute_MSM uses)
        assert compress_blob(native_lib, ctx.msm_host(p96, s32, n)).hex() == case["expected"], case["name"]


def test_empty(native_lib, ctx):
    assert gpu_msm(native_lib, ctx, b"", b"", 0) == bytes([0xC0]) + bytes(47)


@pytest.mark.parametrize("logn,c", [(10, 0), (10, 7), (12, 0), (12, 12), (12, 16), (12, -12), (12, -14), (10, -16), (10, -4)])
def test_seeded_random_vs_naive_oracle(native_lib, ctx, logn, c):
    """Same seeded inputs through the reference algorithm (naive double-and-add loop, C oracle)."""
    rng = random.Random(1000 + logn)
    n = 1 << logn
    base = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(32)]
    # distinct points: base[i % 32] + (i // 32) * G  would cost oracle time; use GPU batch-mul (checked below) instead
    ks = b"".join(rng.randint(1, O.R - 1).to_bytes(32, "little") for _ in range(n))
    dk, dg, dp = ctx.alloc(32 * n), ctx.alloc(96), ctx.alloc(96 * n)
    dk.upload(ks); dg.upload(raw96(O.G1_GEN))
    ctx.batch_mul_device(dg, 1, dk, dp, n)
    p96 = dp.download()
    for i in (0, 1, n // 2, n - 1):   # the generated points themselves are oracle-checked
        assert p96[96 * i: 96 * i + 96] == C.scalar_mul(raw96(O.G1_GEN), ks[32 * i: 32 * i + 32])
    s32 = b"".join(rng.randint(1, O.R - 1).to_bytes(32, "little") for _ in range(n))   # util.py:21-24 distribution
    want = C.compress(C.compute_msm(p96, s32, n))
    assert gpu_msm(native_lib, ctx, p96, s32, n, window_c=c) == want


def test_window_sharding_partials_sum_to_full(native_lib, ctx):
    """The multi-GPU decomposition: partial sums over windows w = rank (mod world) add up to the full MSM."""
    N = native_lib
    rng = random.Random(77)
    n = 300
    pts = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(20)]
    p96 = b"".join(raw96(pts[i % 20]) for i in range(n))
    s32 = b"".join(rng.randint(0, O.R - 1).to_bytes(32, "little") for _ in range(n))
    want = C.compress(C.compute_msm(p96, s32, n))
    dp, ds = ctx.alloc(96 * n), ctx.alloc(32 * n)
    dp.upload(p96); ds.upload(s32)
    for world, c in ((2, 8), (3, 9), (8, 16), (4, 0), (3, -13), (8, -12), (5, -15)):
        cc = c or 10
        acc = ctypes.create_string_buffer(N.POINT_BYTES)
        N.cg1_identity(acc)
        for rank in range(world):
            part = ctx.msm_device(dp, ds, n, window_c=cc, shard_rank=rank, shard_world=world)
            N.cg1_add(acc, acc.raw, part)
        assert compress_blob(N, acc.raw) == want, (world, cc)


@pytest.mark.parametrize("logn", [12, 17])
def test_structured_scalars(native_lib, ctx, logn):
    """Bucket skew: all-equal scalars (same_perm.py:54-55) and sigma = 0..n-1 (curdleproofs.py:315).
    2^17 all-equal terms put 131072 entries into ONE bucket per window: 4096 chunks -> k_heavy_combine."""
    n = 1 << logn
    rng = random.Random(5)
    ks = b"".join(rng.randint(1, O.R - 1).to_bytes(32, "little") for _ in range(n))
    dk, dg, dp = ctx.alloc(32 * n), ctx.alloc(96), ctx.alloc(96 * n)
    dk.upload(ks); dg.upload(raw96(O.G1_GEN))
    ctx.batch_mul_device(dg, 1, dk, dp, n)
    p96 = dp.download()
    k_int = [int.from_bytes(ks[32 * i: 32 * i + 32], "little") for i in range(n)]
    beta = rng.randint(1, O.R - 1)
    two = [(3, O.R - 5)[i & 1] for i in range(n)]                      # two hot buckets per window (+3 and -5's digits)
    few = [rng.randrange(1 << 20) << 16 for _ in range(n)]            # windows 1 and 2 only: window 2 has 16 buckets
    for name, sc in (("all_equal", [beta] * n), ("sigma", list(range(n))), ("all_max", [O.R - 1] * n), ("two_values", two), ("few_bits", few)):
        s32 = b"".join(s.to_bytes(32, "little") for s in sc)
        # closed form: points are k_i * G, so the MSM is (sum k_i s_i mod r) * G
        tot = sum(k * s for k, s in zip(k_int, sc)) % O.R
        want = O.g1_compress(O.g1_mul(O.G1_GEN, tot))
        assert gpu_msm(native_lib, ctx, p96, s32, n) == want, name


def test_full_size_2_20_closed_form_and_linearity(native_lib, ctx):
    """BASELINE metric size.  Points k_i*G (fixed-base batch kernel), scalars uniform mod r:
    MSM == (sum k_i s_i mod r) * G exactly; and MSM(s) + MSM(t) == MSM(s + t)."""
    N = native_lib
    n = 1 << 20
    dk, dg, dp, ds, dt, du = ctx.alloc(32 * n), ctx.alloc(96), ctx.alloc(96 * n), ctx.alloc(32 * n), ctx.alloc(32 * n), ctx.alloc(32 * n)
    ctx.gen_scalars_device(dk, n, 1); ctx.gen_scalars_device(ds, n, 2); ctx.gen_scalars_device(dt, n, 3)
    dg.upload(raw96(O.G1_GEN))
    ctx.batch_mul_device(dg, 1, dk, dp, n)
    kb, sb, tb = dk.download(), ds.download(), dt.download()
    ki = [int.from_bytes(kb[32 * i: 32 * i + 32], "little") for i in range(n)]
    si = [int.from_bytes(sb[32 * i: 32 * i + 32], "little") for i in range(n)]
    ti = [int.from_bytes(tb[32 * i: 32 * i + 32], "little") for i in range(n)]
    assert max(si) < O.R and max(ki) < O.R
    a = ctx.msm_device(dp, ds, n)
    want = O.g1_compress(O.g1_mul(O.G1_GEN, sum(k * s for k, s in zip(ki, si)) % O.R))
    assert compress_blob(N, a) == want
    b = ctx.msm_device(dp, dt, n)
    du.upload(b"".join(((s + t) % O.R).to_bytes(32, "little") for s, t in zip(si, ti)))
    ab = ctypes.create_string_buffer(N.POINT_BYTES)
    N.cg1_add(ab, a, b)
    assert compress_blob(N, ab.raw) == compress_blob(N, ctx.msm_device(dp, du, n))
    # a sample of the generated points against the oracle
    p96 = dp.download(96 * 4)
    for i in range(4):
        assert p96[96 * i: 96 * i + 96] == C.scalar_mul(raw96(O.G1_GEN), kb[32 * i: 32 * i + 32])


def test_config4_full_size_2_22_window_shards(native_lib, ctx):
    """BASELINE config 4: one MSM of 2^22 terms, window buckets sharded over 8 ranks (here: the 8 per-rank partials
    computed one after another on this GPU).  Closed form: points k_i*G, so MSM == (sum k_i s_i mod r) * G exactly;
    the 8 window-shard partials must add up to exactly that point, and so must the 8 point-shard partials and the 8 partials
    of the hybrid split (2 window groups x 4 point groups)."""
    N = native_lib
    n = 1 << 22
    dk, dg, dp, ds = ctx.alloc(32 * n), ctx.alloc(96), ctx.alloc(96 * n), ctx.alloc(32 * n)
    ctx.gen_scalars_device(dk, n, 41); ctx.gen_scalars_device(ds, n, 42)
    dg.upload(raw96(O.G1_GEN))
    ctx.batch_mul_device(dg, 1, dk, dp, n)
    kb, sb = dk.download(), ds.download()
    acc = 0
    for i in range(n):
        acc += int.from_bytes(kb[32 * i: 32 * i + 32], "little") * int.from_bytes(sb[32 * i: 32 * i + 32], "little")
    want = O.g1_compress(O.g1_mul(O.G1_GEN, acc % O.R))
    assert compress_blob(N, ctx.msm_device(dp, ds, n)) == want
    from curdleproofs_pie_amd.distributed import sum_blobs
    parts = [ctx.msm_device(dp, ds, n, window_c=16, shard_rank=g, shard_world=8) for g in range(8)]
    assert compress_blob(N, sum_blobs(parts)) == want
    m = n // 8
    parts = [ctx.msm_device(dp.ptr + 96 * m * g, ds.ptr + 32 * m * g, m) for g in range(8)]
    assert compress_blob(N, sum_blobs(parts)) == want
    # bench.py's default split: 2 window-bucket groups x 4 point groups (distributed.shard_layout)
    from curdleproofs_pie_amd.distributed import shard_layout
    parts = []
    for rank in range(8):
        wr, W, pr, P = shard_layout(rank, 8, "hybrid")
        q = n // P
        parts.append(ctx.msm_device(dp.ptr + 96 * q * pr, ds.ptr + 32 * q * pr, q, window_c=16, shard_rank=wr, shard_world=W))
    assert compress_blob(N, sum_blobs(parts)) == want


def test_two_calls_in_flight(native_lib):
    """cg1_msm_device_begin / _end: two contexts on one GPU hold two MSMs in flight (the next one's sort phases run under this
    one's accumulation); results equal the one-call path, in any interleaving, including n = 0 and a window-sharded call."""
    N = native_lib
    a, b = N.Context(0), N.Context(0)
    try:
        n = 1 << 15
        dk, dg, dp, ds = a.alloc(32 * n), a.alloc(96), a.alloc(96 * n), a.alloc(32 * n)
        a.gen_scalars_device(dk, n, 21); a.gen_scalars_device(ds, n, 22)
        dg.upload(raw96(O.G1_GEN))
        a.batch_mul_device(dg, 1, dk, dp, n)
        sizes = [n, 1000, n // 2, 0, 7, n]
        want = [a.msm_device(dp, ds, m) for m in sizes]
        got, pending = [], None
        for k, m in enumerate(sizes):
            cx = (a, b)[k % 2]
            cx.msm_device_begin(dp, ds, m)
            if pending is not None:
                got.append(pending.msm_device_end())
            pending = cx
        got.append(pending.msm_device_end())
        assert all(N.cg1_eq(g, w) == 1 for g, w in zip(got, want)) and len(got) == len(want)
        from curdleproofs_pie_amd.distributed import sum_blobs

        parts, got4 = [], []
        for rank in range(4):                                   # the four window shards of one MSM, two in flight at a time
            cx = (a, b)[rank % 2]
            cx.msm_device_begin(dp, ds, n, window_c=16, shard_rank=rank, shard_world=4)
            parts.append(cx)
            if len(parts) == 2:
                got4 += [p.msm_device_end() for p in parts]
                parts = []
        assert compress_blob(N, sum_blobs(got4)) == compress_blob(N, want[0])
    finally:
        a.close(); b.close()


def test_batch_mul_variable_base(native_lib, ctx):
    rng = random.Random(9)
    n = 37
    bases = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(5)] + [None]
    sc = [rng.randint(0, O.R - 1) for _ in range(n)]
    sc[3] = 0
    db, ds, do = ctx.alloc(96 * 6), ctx.alloc(32 * n), ctx.alloc(96 * n)
    db.upload(b"".join(raw96(p) for p in bases)); ds.upload(b"".join(s.to_bytes(32, "little") for s in sc))
    ctx.batch_mul_device(db, 6, ds, do, n)
    out = do.download()
    for i in range(n):
        assert out[96 * i: 96 * i + 96] == raw96(O.g1_mul(bases[i % 6], sc[i])), i


def test_batch_mul_add_pool_equals_device(native_lib, ctx):
    """cg1_batch_mul_add with host pointers: the worker pool's path (few outputs) and the GPU kernel give the same records, byte for byte
    (fold, same-scalar map and per-index patterns; identity bases / addends; scalar 0 and r - 1)."""
    import ctypes

    N = native_lib
    rng = random.Random(77)
    n = 41
    pts = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(n)]
    pts[4] = None
    add = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(n)]
    add[9] = None
    add[11] = O.g1_neg(pts[11])
    sc = [rng.randint(0, O.R - 1) for _ in range(n)]
    sc[3], sc[11], sc[12] = 0, 1, O.R - 1
    b96, a96 = b"".join(raw96(p) for p in pts), b"".join(raw96(p) for p in add)
    s32 = b"".join(s.to_bytes(32, "little") for s in sc)
    for nbase, scal, nsc, addend in ((n, s32, n, None), (n, s32[:32 * 5], 5, a96), (n, s32[32 * 20: 32 * 21], 1, a96), (1, s32, n, None), (6, s32, n, a96)):
        outs = []
        for host_max in (0, 1 << 20):
            ctx.set_param("batch_mul_host_max", host_max)
            try:
                outs.append(ctx.batch_mul_add_host(b96, nbase, scal, nsc, addend, n))
            finally:
                ctx.set_param("batch_mul_host_max", -1)
        assert outs[0] == outs[1]
        for i in range(n):
            sci = int.from_bytes(scal[32 * (i % nsc): 32 * (i % nsc) + 32], "little")
            want = O.g1_add(None if addend is None else add[i], O.g1_mul(pts[i % nbase], sci))
            assert outs[0][96 * i: 96 * i + 96] == raw96(want), i


def test_batched_small_msms_ragged(native_lib, ctx):
    """Regime B: independent MSMs of ragged sizes (incl. empty) in one launch chain, each vs the naive oracle."""
    N = native_lib
    rng = random.Random(31)
    base = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(40)] + [None]
    sizes = [0, 1, 2, 5, 0, 64, 131, 307, 627, 3, 33]
    pts, sc, offsets = [], [], [0]
    for n in sizes:
        for _ in range(n):
            pts.append(base[rng.randrange(len(base))])
            sc.append(rng.choice([0, 1, O.R - 1, rng.randint(0, O.R - 1), rng.randint(0, O.R - 1)]))
        offsets.append(len(pts))
    p96 = b"".join(raw96(p) for p in pts)
    s32 = b"".join(s.to_bytes(32, "little") for s in sc)
    for c in (0, 4, 6, 9):
        dp, ds = ctx.alloc(len(p96)), ctx.alloc(len(s32))
        dp.upload(p96); ds.upload(s32)
        blobs = ctx.msm_batched_device(dp, ds, offsets, window_c=c)
        assert len(blobs) == len(sizes)
        for j, n in enumerate(sizes):
            lo, hi = offsets[j], offsets[j + 1]
            want = C.compress(C.compute_msm(p96[96 * lo: 96 * hi], s32[32 * lo: 32 * hi], n))
            assert compress_blob(N, blobs[j]) == want, (c, j, n)
    assert [compress_blob(N, b) for b in ctx.msm_batched_host(p96, s32, offsets)] == [compress_blob(N, b) for b in blobs]


def test_batched_1024_x_627_closed_form(native_lib, ctx):
    """BASELINE config 3's MSM content: 1024 independent 627-term MSMs (5*ell+7 at ell=124).  Points k_i*G,
    so MSM_j == (sum_{i in j} k_i s_i mod r) * G exactly."""
    N = native_lib
    M, n = 1024, 627
    tot = M * n
    dk, dg, dp, ds = ctx.alloc(32 * tot), ctx.alloc(96), ctx.alloc(96 * tot), ctx.alloc(32 * tot)
    ctx.gen_scalars_device(dk, tot, 11); ctx.gen_scalars_device(ds, tot, 12)
    dg.upload(raw96(O.G1_GEN))
    ctx.batch_mul_device(dg, 1, dk, dp, tot)
    kb, sb = dk.download(), ds.download()
    blobs = ctx.msm_batched_device(dp, ds, [n * j for j in range(M + 1)])
    assert len(blobs) == M
    for j in list(range(0, M, 97)) + [M - 1]:
        acc = 0
        for i in range(n * j, n * (j + 1)):
            acc += int.from_bytes(kb[32 * i: 32 * i + 32], "little") * int.from_bytes(sb[32 * i: 32 * i + 32], "little")
        assert compress_blob(N, blobs[j]) == O.g1_compress(O.g1_mul(O.G1_GEN, acc % O.R)), j
    # and every MSM of the batch agrees with the single-MSM path (regime A kernels)
    for j in (0, 511, 1023):
        single = ctx.msm_device(dp.ptr + 96 * n * j, ds.ptr + 32 * n * j, n)
        assert N.cg1_eq(single, blobs[j]) == 1


def test_pipeline_variants_agree(native_lib, golden):
    """Same results through the alternative code paths: global-atomic counting sort (used for n > 2^23),
    other chunk lengths / segment sizes, wave-aggregation off."""
    N = native_lib
    c2 = N.Context(0)
    try:
        rng = random.Random(55)
        n = 3000
        base = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(24)]
        p96 = b"".join(raw96(base[rng.randrange(24)]) for _ in range(n))
        s32 = b"".join(rng.choice([rng.randint(0, O.R - 1), 7, O.R - 1]).to_bytes(32, "little") for _ in range(n))
        want = C.compress(C.compute_msm(p96, s32, n))
        for params in ({"partition_sort": 0}, {"chunk_len": 1}, {"chunk_len": 7}, {"chunk_len": 4096}, {"seg_m": 1},
                       {"seg_m": 2}, {"seg_m": 16}, {"wave_agg": 0}, {"stage_sort": 0}, {"host_split": 0}, {"big_bins": 0}, {"partition_sort": 0, "chunk_len": 3, "seg_m": 8},
                       {"reduce_2d": 0}, {"reduce_2d": 0, "seg_m": 8}, {"quad": 0}, {"rowcol_quad": 0}, {"auto_plan": 0}):
            for k, v in params.items():
                c2.set_param(k, v)
            for c in (0, 4, 5, 6, 9, 16, -10, -13):
                assert gpu_msm(N, c2, p96, s32, n, window_c=c) == want, (params, c)
            for k in params:   # back to defaults
                c2.set_param(k, {"partition_sort": 1, "chunk_len": 8, "seg_m": 4, "wave_agg": 1, "stage_sort": 1, "host_split": 1, "big_bins": 1, "reduce_2d": 1, "quad": 1, "rowcol_quad": 1, "auto_plan": 1}[k])
    finally:
        c2.close()


def test_identity_terms_contribute_nothing(native_lib, ctx):
    """Many identity bases ((0, 0) records) among few distinct scalars: whole buckets of identities, identities first,
    second and last in a bucket."""
    N = native_lib
    rng = random.Random(4242)
    base = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(6)]
    for n, p_inf, few_scalars in ((64, 0.5, True), (700, 0.7, True), (5000, 0.3, False), (3000, 1.0, True)):
        pts = [None if rng.random() < p_inf else base[rng.randrange(6)] for _ in range(n)]
        pool = [rng.randint(1, O.R - 1) for _ in range(3)]          # few distinct scalars: long chunks with many identity entries
        sc = [rng.choice(pool) if few_scalars else rng.randint(0, O.R - 1) for _ in range(n)]
        p96 = b"".join(raw96(p) for p in pts)
        s32 = b"".join(s.to_bytes(32, "little") for s in sc)
        want = C.compress(C.msm_bucket(p96, s32, n))
        for c in (0, 8, 13, -12, 16):
            assert gpu_msm(N, ctx, p96, s32, n, window_c=c) == want, (n, p_inf, c)


def test_two_contexts_concurrently(native_lib):
    """One context per stream: independent MSMs issued from two host threads on the same GPU."""
    import threading

    N = native_lib
    rng = random.Random(66)
    n = 2000
    base = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(16)]
    jobs = []
    for _ in range(2):
        p96 = b"".join(raw96(base[rng.randrange(16)]) for _ in range(n))
        s32 = b"".join(rng.randint(0, O.R - 1).to_bytes(32, "little") for _ in range(n))
        jobs.append((p96, s32, C.compress(C.compute_msm(p96, s32, n))))
    ctxs = [N.Context(0), N.Context(0)]
    out = [None, None]

    def work(i):
        res = []
        for _ in range(5):
            res.append(compress_blob(N, ctxs[i].msm_host(jobs[i][0], jobs[i][1], n)))
        out[i] = res

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i in range(2):
        assert out[i] == [jobs[i][2]] * 5
    for c in ctxs:
        c.close()


@pytest.mark.parametrize("logn", [16, 18])
def test_config2_full_size_vs_cpu_oracle(native_lib, ctx, logn):
    """BASELINE config 2: one MSM of 2^16 random G1 points, bit-exact vs the CPU (and 2^18 for good measure).
    CPU side: the C oracle's bucket method (cross-checked against its naive reference loop on CPU)."""
    n = 1 << logn
    dk, dg, dp, ds = ctx.alloc(32 * n), ctx.alloc(96), ctx.alloc(96 * n), ctx.alloc(32 * n)
    ctx.gen_scalars_device(dk, n, 100 + logn); ctx.gen_scalars_device(ds, n, 200 + logn)
    dg.upload(raw96(O.G1_GEN))
    ctx.batch_mul_device(dg, 1, dk, dp, n)
    p96, s32 = dp.download(), ds.download()
    want = C.compress(C.msm_bucket(p96, s32, n))
    for c in (0, 16):
        assert compress_blob(native_lib, ctx.msm_device(dp, ds, n, window_c=c)) == want


def test_argument_errors_are_reported(native_lib, ctx):
    """Bad arguments come back as status codes -> NativeError with a message; nothing is silently 'fixed'."""
    N = native_lib
    dp, ds = ctx.alloc(96 * 4), ctx.alloc(32 * 4)
    dp.upload(raw96(O.G1_GEN) * 4); ds.upload((5).to_bytes(32, "little") * 4)
    for kw in ({"window_c": 3}, {"window_c": 17}, {"window_c": -3}, {"window_c": -17}, {"shard_rank": 2, "shard_world": 2}, {"shard_rank": -1, "shard_world": 1},
               {"shard_world": 0}):
        with pytest.raises(N.NativeError):
            ctx.msm_device(dp, ds, 4, **kw)
    with pytest.raises(N.NativeError):
        ctx.msm_batched_device(dp, ds, [0, 3, 2, 4])           # offsets not monotone
    with pytest.raises(N.NativeError):
        ctx.msm_batched_device(dp, ds, [1, 4])                 # offsets[0] != 0
    with pytest.raises(N.NativeError):
        ctx.msm_batched_device(dp, ds, [0, 4], window_c=12)    # batched widths are 4..9
    with pytest.raises(N.NativeError):
        ctx.set_param("no_such_param", 1)
    # scalars must fit the signed-digit recoding (< 2^255; canonical Fr elements always do): rejected, not mis-summed
    dbad = ctx.alloc(32 * 4)
    dbad.upload((5).to_bytes(32, "little") * 3 + (1 << 255).to_bytes(32, "little"))
    with pytest.raises(N.NativeError, match="2\\^255"):
        ctx.msm_device(dp, dbad, 4)
    with pytest.raises(N.NativeError, match="2\\^255"):
        ctx.msm_batched_device(dp, dbad, [0, 2, 4])
    dok = ctx.alloc(32 * 4)
    dok.upload((5).to_bytes(32, "little") * 3 + ((1 << 255) - 1).to_bytes(32, "little"))     # non-canonical but in range: plain integer
    assert compress_blob(N, ctx.msm_device(dp, dok, 4)) == O.g1_compress(O.g1_mul(O.G1_GEN, (15 + (1 << 255) - 1) % O.R))
    # the context stays usable after errors
    want = O.g1_compress(O.g1_mul(O.G1_GEN, 20))
    assert compress_blob(N, ctx.msm_device(dp, ds, 4)) == want
    # ranks beyond the window count own nothing: identity partial
    assert compress_blob(N, ctx.msm_device(dp, ds, 4, window_c=16, shard_rank=17, shard_world=20)) == bytes([0xC0]) + bytes(47)


def test_randomised_differential(native_lib, ctx):
    """A few hundred random (n, window width, shard, input mix) cases against the C oracle's bucket MSM:
    boundary sizes of the sort tiles (4096) and scan blocks (1024), identity bases, duplicates, negated pairs,
    zero / one / r-1 / power-of-two / small scalars."""
    N = native_lib
    rng = random.Random(20241003)
    base = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(48)]
    base += [O.g1_neg(p) for p in base[:8]] + [None]
    raws = [raw96(p) for p in base]
    sizes = [1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192, 8193, 12289]
    checked = 0
    for it in range(160):
        n = rng.choice(sizes) if it % 3 else rng.randint(1, 3000)
        kind = rng.randrange(5)
        idx = [rng.randrange(len(raws)) if kind != 1 else rng.randrange(4) for _ in range(n)]
        def scalar():
            r = rng.random()
            if kind == 2:
                return rng.choice([0, 1, O.R - 1, 1 << rng.randrange(255), (1 << rng.randrange(1, 255)) - 1])
            if kind == 3:
                return rng.randrange(1 << 20)
            return 0 if r < 0.02 else rng.randint(0, O.R - 1)
        sc = [scalar() for _ in range(n)]
        if kind == 4:
            sc = [sc[0]] * n
        p96 = b"".join(raws[i] for i in idx)
        s32 = b"".join(s.to_bytes(32, "little") for s in sc)
        want = C.compress(C.msm_bucket(p96, s32, n))
        c = rng.choice([0, 0, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, -4, -6, -9, -11, -12, -13, -14, -15, -16])
        dp, ds = ctx.alloc(96 * n), ctx.alloc(32 * n)
        dp.upload(p96); ds.upload(s32)
        if rng.random() < 0.3:
            world = rng.choice([2, 3, 5, 8])
            cc = c or 9
            acc = ctypes.create_string_buffer(N.POINT_BYTES)
            N.cg1_identity(acc)
            for rk in range(world):
                N.cg1_add(acc, acc.raw, ctx.msm_device(dp, ds, n, window_c=cc, shard_rank=rk, shard_world=world))
            got = compress_blob(N, acc.raw)
        else:
            got = compress_blob(N, ctx.msm_device(dp, ds, n, window_c=c))
        dp.free(); ds.free()
        assert got == want, (it, n, kind, c)
        checked += 1
    assert checked == 160


def test_randomised_batched_differential(native_lib, ctx):
    N = native_lib
    rng = random.Random(77001)
    base = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(32)] + [None]
    raws = [raw96(p) for p in base]
    for it in range(12):
        M = rng.randint(1, 40)
        sizes = [rng.choice([0, 1, 2, 7, 64, 255, 256, 257, 700]) for _ in range(M)]
        offs, p, s = [0], [], []
        for n in sizes:
            for _ in range(n):
                p.append(raws[rng.randrange(len(raws))])
                s.append(rng.choice([0, 1, O.R - 1, rng.randint(0, O.R - 1), rng.randint(0, O.R - 1)]).to_bytes(32, "little"))
            offs.append(offs[-1] + n)
        p96, s32 = b"".join(p), b"".join(s)
        if not p96:
            continue
        dp, ds = ctx.alloc(len(p96)), ctx.alloc(len(s32))
        dp.upload(p96); ds.upload(s32)
        blobs = ctx.msm_batched_device(dp, ds, offs, window_c=rng.choice([0, 4, 5, 6, 7, 8, 9]))
        dp.free(); ds.free()
        for j, n in enumerate(sizes):
            want = C.compress(C.msm_bucket(p96[96 * offs[j]: 96 * offs[j + 1]], s32[32 * offs[j]: 32 * offs[j + 1]], n, 6))
            assert compress_blob(N, blobs[j]) == want, (it, j, n)


def test_two_chain_split_equals_single_chain(native_lib, ctx):
    """The two-chain form of a large call ("split": high / low half of its windows on two streams; an A/B switch that stays off, it
    measured slower): same point as the single chain, for whole MSMs and for the window shares of a sharded one (the halves are then halves of the share), at the automatic
    plan, a uniform and a balanced one; the closed form pins the value."""
    N = native_lib
    n = (1 << 17) + 333
    rng = random.Random(91)
    dk, dg, dp, ds = ctx.alloc(32 * n), ctx.alloc(96), ctx.alloc(96 * n), ctx.alloc(32 * n)
    dg.upload(raw96(O.G1_GEN))
    ctx.gen_scalars_device(dk, n, 91)
    ctx.batch_mul_device(dg, 1, dk, dp, n)
    ctx.gen_scalars_device(ds, n, 92)
    ks, sc = dk.download(), ds.download()
    tot = sum(int.from_bytes(ks[32 * i: 32 * i + 32], "little") * int.from_bytes(sc[32 * i: 32 * i + 32], "little") for i in range(n)) % O.R
    want = O.g1_compress(O.g1_mul(O.G1_GEN, tot))
    try:
        for c in (0, 16, -13, 9):
            ctx.set_param("split", 1)
            ctx.set_param("split_min_log2n", 10)
            got = ctx.msm_device(dp, ds, n, window_c=c)
            assert ctx.last_counts()["accumulate_launches"] == 2
            assert compress_blob(N, got) == want, c
            ctx.set_param("split", 0)
            single = ctx.msm_device(dp, ds, n, window_c=c)
            assert ctx.last_counts()["accumulate_launches"] == 1 and N.cg1_eq(single, got) == 1
        for world in (2, 3, 8):
            for split in (1, 0):
                ctx.set_param("split", split)
                acc = ctypes.create_string_buffer(N.POINT_BYTES)
                N.cg1_identity(acc)
                for rank in range(world):
                    N.cg1_add(acc, acc.raw, ctx.msm_device(dp, ds, n, window_c=16, shard_rank=rank, shard_world=world))
                assert compress_blob(N, acc.raw) == want, (world, split)
    finally:
        ctx.set_param("split", 0)
        ctx.set_param("split_min_log2n", 17)
    for b in (dk, dg, dp, ds):
        b.free()
