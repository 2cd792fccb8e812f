"""k_msm_small -- the single-launch MSM of the protocol's own sizes (4 ... 627 terms: ipa.py:223,232; same_msm.py:219-226;
msm_accumulator.py:64; anything up to 2048) -- against the CPU oracle, bit for bit, through the C ABI.  Needs an MI355X.

Covers every size 1 ... 1024, every window width the kernel runs (4, 6, 7, 8, 9), all four point sources (affine96, normalised
blobs, projective blobs, resident prepared records), the callers' skewed scalar patterns (all-equal, sigma = 0..n-1, r - 1,
zeros), identity / duplicate / opposite bases, and the A/B switch back to the regime-A chain."""
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


@pytest.fixture(scope="module")
def pool():
    rng = random.Random(404)
    ks = [rng.randint(1, O.R - 1) for _ in range(96)]
    return ks, [O.g1_mul(O.G1_GEN, k) for k in ks]


def compress_blob(N, blob):
    out = ctypes.create_string_buffer(48)
    N.cg1_compress(out, blob)
    return out.raw


def test_every_size_1_to_1024(native_lib, ctx, pool):
    """Random bases (drawn with repetition from 96 oracle-made points, an identity now and then) and uniform scalars at every n."""
    N = native_lib
    ks, pts = pool
    rng = random.Random(405)
    for n in range(1, 1025):
        idx = [rng.randrange(len(pts)) for _ in range(n)]
        p96 = b"".join(bytes(96) if rng.random() < 0.01 else raw96(pts[i]) for i in idx)
        s32 = b"".join(rng.randint(0, O.R - 1).to_bytes(32, "little") for _ in range(n))
        want = C.compress(C.msm_bucket(p96, s32, n))
        assert compress_blob(N, ctx.msm_host(p96, s32, n)) == want, n
    assert ctx.timings()["window_c"] == 7                       # the last call (n = 1024) ran the small kernel (its plan for n > 96)


def test_sizes_up_to_2048_tree_combine(native_lib, ctx, pool):
    """5 ... 8 slices per window: the last workgroup of a window adds the slices' items as a tree over pairs (every slice count, sizes on
    both sides of every slice edge; c = 7 up to 1 536 terms, c = 8 up to 2 048 = one round of workgroups; 2 049 takes the launch chain)."""
    N = native_lib
    ks, pts = pool
    rng = random.Random(415)
    sizes = [1025, 1279, 1280, 1281] + [256 * s + d for s in range(5, 8) for d in (0, 1)] + [1391, 1535, 1536, 1537, 2000, 2047, 2048]
    for n in sorted(set(sizes)):
        idx = [rng.randrange(len(pts)) for _ in range(n)]
        sc = [rng.randint(0, O.R - 1) for _ in range(n)]
        p96 = b"".join(bytes(96) if rng.random() < 0.01 else raw96(pts[i]) for i in idx)
        s32 = b"".join(v.to_bytes(32, "little") for v in sc)
        want = C.compress(C.msm_bucket(p96, s32, n))
        assert compress_blob(N, ctx.msm_host(p96, s32, n)) == want, n
        assert ctx.last_counts()["accumulate_launches"] == 0, n          # k_msm_small served it
        assert ctx.timings()["window_c"] == (7 if n <= 1536 else 8), n
        if n in (1281, 1536, 2048):
            dp, ds = ctx.alloc(96 * n), ctx.alloc(32 * n)
            dp.upload(p96); ds.upload(s32)
            for c in (6, 7, 8, 9):                                     # other widths: other numbers of items per window in the tree
                assert compress_blob(N, ctx.msm_device(dp, ds, n, window_c=c)) == want, (n, c)
            dp.free(); ds.free()
            s_eq = (sc[0].to_bytes(32, "little")) * n                   # every term in ONE bucket per window and slice
            assert compress_blob(N, ctx.msm_host(p96, s_eq, n)) == C.compress(C.msm_bucket(p96, s_eq, n)), n
    n = 2049
    idx = [rng.randrange(len(pts)) for _ in range(n)]
    p96 = b"".join(raw96(pts[i]) for i in idx)
    s32 = b"".join(rng.randint(0, O.R - 1).to_bytes(32, "little") for _ in range(n))
    assert compress_blob(N, ctx.msm_host(p96, s32, n)) == C.compress(C.msm_bucket(p96, s32, n))
    assert ctx.last_counts()["accumulate_launches"] == 1


@pytest.mark.parametrize("c", [4, 6, 7, 8, 9])
def test_every_window_width(native_lib, ctx, pool, c):
    N = native_lib
    ks, pts = pool
    rng = random.Random(406 + c)
    for n in (1, 2, 3, 7, 64, 255, 256, 257, 300, 512, 513, 627, 768, 769, 1023, 1024):
        idx = [rng.randrange(len(pts)) for _ in range(n)]
        p96 = b"".join(raw96(pts[i]) for i in idx)
        sc = [rng.randint(0, O.R - 1) for _ in range(n)]
        s32 = b"".join(v.to_bytes(32, "little") for v in sc)
        dp, ds = ctx.alloc(96 * n), ctx.alloc(32 * n)
        dp.upload(p96); ds.upload(s32)
        got = compress_blob(N, ctx.msm_device(dp, ds, n, window_c=c))
        assert ctx.timings()["window_c"] == c
        tot = sum(ks[i] * s for i, s in zip(idx, sc)) % O.R
        assert got == O.g1_compress(O.g1_mul(O.G1_GEN, tot)), (c, n)
        dp.free(); ds.free()


def test_structured_scalars_and_exceptional_bases(native_lib, ctx, pool):
    """The callers' patterns: [beta] * ell (same_perm.py:54-55) puts every term of a window into ONE bucket -- n / 4 chunks joined by
    the fold tree; sigma = 0..n-1 (curdleproofs.py:315) leaves the high windows empty; duplicate bases, P and -P, identities
    (curdleproofs.py:124-136 really pass these)."""
    N = native_lib
    ks, pts = pool
    rng = random.Random(407)
    neg = lambda p: (p[0], O.P - p[1])
    for n in (5, 124, 128, 307, 627, 1024):
        idx = [rng.randrange(len(pts)) for _ in range(n)]
        base_pts = [pts[i] for i in idx]
        beta = rng.randint(1, O.R - 1)
        patterns = {
            "all_equal": [beta] * n,
            "sigma": list(range(n)),
            "all_r_minus_1": [O.R - 1] * n,
            "all_zero": [0] * n,
            "one_hot": [0] * (n - 1) + [beta],
            "two_values": [(3, O.R - 5)[i & 1] for i in range(n)],
            "top_bits": [(1 << 254) + i for i in range(n)],
        }
        p96 = b"".join(raw96(p) for p in base_pts)
        for name, sc in patterns.items():
            s32 = b"".join(v.to_bytes(32, "little") for v in sc)
            tot = sum(ks[i] * s for i, s in zip(idx, sc)) % O.R
            assert compress_blob(N, ctx.msm_host(p96, s32, n)) == O.g1_compress(O.g1_mul(O.G1_GEN, tot)), (n, name)
        # all bases equal (doubling inside every chunk), P / -P pairs (cancellation), identities in between
        same = [base_pts[0]] * n
        pm = [base_pts[0] if i & 1 else neg(base_pts[0]) for i in range(n)]
        mixed = [None if i % 3 == 0 else base_pts[i] for i in range(n)]
        sc = [rng.randint(0, O.R - 1) for _ in range(n)]
        s32 = b"".join(v.to_bytes(32, "little") for v in sc)
        one = b"".join((1).to_bytes(32, "little") for _ in range(n))
        for name, bl, s in (("same_base", same, s32), ("same_base_same_scalar", same, one), ("plus_minus", pm, one), ("plus_minus_random", pm, s32), ("identities", mixed, s32)):
            p = b"".join(raw96(x) for x in bl)
            assert compress_blob(N, ctx.msm_host(p, s, n)) == C.compress(C.msm_bucket(p, s, n)), (n, name)


def test_scalar_out_of_range_is_rejected(native_lib, ctx, pool):
    N = native_lib
    ks, pts = pool
    n = 300
    p96 = b"".join(raw96(pts[i % len(pts)]) for i in range(n))
    sc = [5] * n
    sc[123] = 1 << 255
    with pytest.raises(N.NativeError):
        ctx.msm_host(p96, b"".join(v.to_bytes(32, "little") for v in sc), n)
    # and the context keeps working (the kernel's counters were left clean)
    s32 = b"".join((7).to_bytes(32, "little") for _ in range(n))
    tot = sum(ks[i % len(pts)] * 7 for i in range(n)) % O.R
    assert compress_blob(N, ctx.msm_host(p96, s32, n)) == O.g1_compress(O.g1_mul(O.G1_GEN, tot))


@pytest.mark.parametrize("glv", [0, 1])
def test_point_sources_and_ab_switch(native_lib, ctx, pool, glv):
    """affine96, blobs (Z = 1 and projective), resident prepared records: the same result; `small_msm` = 0 sends the call through
    the regime-A launch chain, which must agree too.  glv = 1: the same with the endomorphism split (the pool's points are multiples
    of the generator, so the promise the parameter makes holds)."""
    N = native_lib
    ks, pts = pool
    rng = random.Random(408)
    g = ctypes.create_string_buffer(144)
    N.cg1_generator(g)
    ctx.set_param("glv", glv)
    for n in (3, 200, 627, 1000):
        idx = [rng.randrange(len(pts)) for _ in range(n)]
        sc = [rng.randint(0, O.R - 1) for _ in range(n)]
        s32 = b"".join(v.to_bytes(32, "little") for v in sc)
        p96 = b"".join(raw96(pts[i]) for i in idx)
        want = O.g1_compress(O.g1_mul(O.G1_GEN, sum(ks[i] * s for i, s in zip(idx, sc)) % O.R))
        assert compress_blob(N, ctx.msm_host(p96, s32, n)) == want
        norm = ctypes.create_string_buffer(144 * n)
        assert N.cg1_batch_from_affine96(norm, p96, n) == N.OK
        assert compress_blob(N, ctx.msm_blobs(norm.raw, s32, n, True)) == want
        proj = []
        for i in idx:                                                           # k_i * G by the host ladder: Z != 1
            out = ctypes.create_string_buffer(144)
            N.cg1_mul(out, g.raw, ks[i].to_bytes(32, "little"))
            proj.append(out.raw)
        proj = b"".join(proj)
        assert compress_blob(N, ctx.msm_blobs(proj, s32, n, False)) == want
        vec = ctx.vec(proj, n, False)
        assert compress_blob(N, ctx.msm_vec(vec, s32, n)) == want
        if n > 10:
            first, m = 7, n - 10
            part = O.g1_compress(O.g1_mul(O.G1_GEN, sum(ks[idx[i]] * sc[i] for i in range(first, first + m)) % O.R))
            assert compress_blob(N, ctx.msm_vec(vec, s32[32 * first: 32 * (first + m)], m, first)) == part
        vec.free()
        ctx.set_param("small_msm", 0)
        try:
            assert compress_blob(N, ctx.msm_host(p96, s32, n)) == want
            assert ctx.timings()["window_c"] in (4, 8)                          # the regime-A plan for these sizes
        finally:
            ctx.set_param("small_msm", 1)
    ctx.set_param("glv", 0)


def test_golden_vectors_through_the_small_kernel(native_lib, ctx, golden):
    """tests/golden/msm_vectors.json (23 oracle-made cases incl. the edge cases of SURVEY 8(c)) at every width of the small kernel."""
    N = native_lib
    for case in golden:
        pts = [O.g1_decompress(bytes.fromhex(h)) for h in case["points"]]
        n = len(pts)
        if n == 0 or n > 2048:
            continue
        p96 = b"".join(raw96(p) for p in pts)
        s32 = b"".join(bytes.fromhex(h) for h in case["scalars"])
        dp, ds = ctx.alloc(96 * n), ctx.alloc(32 * n)
        dp.upload(p96); ds.upload(s32)
        for c in (0, 4, 6, 7, 8, 9):
            assert compress_blob(N, ctx.msm_device(dp, ds, n, window_c=c)).hex() == case["expected"], (case["name"], c)
        dp.free(); ds.free()


@pytest.mark.parametrize("glv", [0, 1])
def test_several_msms_in_one_launch(native_lib, ctx, pool, glv):
    """cg1_msm_batched with a handful of small MSMs (the 4 - 6 of a prover's halving round; compute_MSM_batch): they ride ONE
    k_msm_small launch (grid.z = MSM).  Ragged sizes, an empty MSM in the middle, sizes across the slice edge; M = 17 falls back to
    the regime-B launch chain -- every result against the oracle."""
    N = native_lib
    ks, pts = pool
    rng = random.Random(409)
    ctx.set_param("glv", glv)
    for sizes in ([5], [3, 0, 7], [64, 65, 64, 64], [128, 129, 1, 128, 0, 127], [257, 300, 2], [1024, 1], [1500, 700], [2048], [33] * 16, [20] * 17, [0, 0, 9]):
        idx = [[rng.randrange(len(pts)) for _ in range(n)] for n in sizes]
        sc = [[rng.randint(0, O.R - 1) for _ in range(n)] for n in sizes]
        p96 = b"".join(raw96(pts[i]) for row in idx for i in row)
        s32 = b"".join(v.to_bytes(32, "little") for row in sc for v in row)
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        blobs = ctx.msm_batched_host(p96, s32, offs)
        assert len(blobs) == len(sizes)
        for j, n in enumerate(sizes):
            tot = sum(ks[i] * s for i, s in zip(idx[j], sc[j])) % O.R
            assert compress_blob(N, blobs[j]) == O.g1_compress(O.g1_mul(O.G1_GEN, tot)), (sizes, j)
        if len(sizes) <= 16 and max(sizes) <= 2048:
            assert ctx.last_counts()["accumulate_launches"] == 0              # k_msm_small served it
    # an out-of-range scalar in one of the MSMs rejects the call and leaves the counters clean
    bad = bytearray(s32); bad[31] |= 0x80
    with pytest.raises(N.NativeError):
        ctx.msm_batched_host(p96, bytes(bad), offs)
    assert compress_blob(N, ctx.msm_batched_host(p96, s32, offs)[2]) == O.g1_compress(O.g1_mul(O.G1_GEN, sum(ks[i] * s for i, s in zip(idx[2], sc[2])) % O.R))
    ctx.set_param("glv", 0)
