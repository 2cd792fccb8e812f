"""compute_MSM / MSMAccumulator / multiexp_unchecked with the reference's signatures, on the GPU.
Reads like the reference's own tests (test_curdleproofs.py) for this path."""
import random

import pytest

from oracle import bls12_381 as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["deferred", "eager"])
def api(native_lib, request):
    """Every test of this module runs with the operators deferred (the default: values evaluated in batches when bytes / comparisons are
    asked for) and computing at once (CURDLE_G1_LAZY=0): the same results either way."""
    import curdleproofs_pie_amd as A
    import curdleproofs_pie_amd.py_arkworks_bls12381 as B
    from curdleproofs_pie_amd import util as U

    prev = B.set_lazy(request.param == "deferred")
    yield A, U
    B.set_lazy(prev)


def naive(bases, scalars, A):
    cur = A.G1Point.identity()                      # msm_accumulator.py:9-12 with the host operators
    for b, s in zip(bases, scalars):
        cur = cur + b * s
    return cur


def test_compute_msm_matches_reference_loop(api):
    A, U = api
    random.seed(3)
    for n in (0, 1, 4, 7, 131, 627):
        bases = [U.get_random_point() for _ in range(n)]
        scalars = [U.random_scalar() for _ in range(n)]
        got = A.compute_MSM(bases, scalars)
        assert got == naive(bases, scalars, A)
        want = O.compute_MSM_fast([O.g1_decompress(b.to_compressed_bytes()) for b in bases], [int(s) for s in scalars]) if n else None
        assert bytes(got.to_compressed_bytes()) == O.g1_compress(want)
    # zip semantics: iterators, truncation to the shorter (test_curdleproofs.py:432 passes a map object)
    bases = [U.get_random_point() for _ in range(5)]
    scalars = [U.random_scalar() for _ in range(3)]
    assert A.compute_MSM(iter(bases), map(lambda s: s, scalars)) == naive(bases[:3], scalars, A)
    assert A.G1Point.multiexp_unchecked(bases[:3], scalars) == naive(bases[:3], scalars, A)
    # identity bases and zero scalars (curdleproofs.py:74, :124-136 really pass these)
    z = [U.Z1, bases[0], bases[1]]
    s = [scalars[0], A.Scalar(0), scalars[1]]
    assert A.compute_MSM(z, s) == bases[1] * scalars[1]
    assert A.compute_MSM([U.Z1], [scalars[0]]) == U.Z1


def test_msm_accumulator(api):
    A, U = api
    random.seed(4)
    G = [U.get_random_point() for _ in range(16)]
    a = [U.random_scalar() for _ in range(16)]
    b = [U.random_scalar() for _ in range(9)]
    acc = A.MSMAccumulator()
    acc.accumulate_check(naive(G, a, A), G + [U.Z1], a + [A.Scalar(5)])
    acc.accumulate_check(naive(G[:9], b, A), G[:9], b)
    assert len(acc.base_scalar_map) == 16            # equal bases merged (:54-58), identity skipped (:49-50)
    acc.verify()
    bad = A.MSMAccumulator()
    bad.accumulate_check(naive(G, a, A) + U.G1, G, a)
    bad.accumulate_check(naive(G[:9], b, A), G[:9], b)
    with pytest.raises(AssertionError):              # :68
        bad.verify()
    with pytest.raises(ValueError):                  # :63 on an empty map
        A.MSMAccumulator().verify()
    # exactly one random_scalar() draw per accumulate_check, from Python's global `random` (:43)
    random.seed(123); acc2 = A.MSMAccumulator(); acc2.accumulate_check(naive(G, a, A), G, a); after = random.random()
    random.seed(123); U.random_scalar(); assert random.random() == after
    assert acc2.A_c == naive(G, a, A) * A.Scalar(acc2._lhs[0][1])


def test_compute_msm_batch_and_verify_many(api):
    A, U = api
    from curdleproofs_pie_amd.msm_accumulator import compute_MSM_batch

    random.seed(6)
    jobs = []
    for n in (3, 0, 17, 64):
        jobs.append(([U.get_random_point() for _ in range(n)], [U.random_scalar() for _ in range(n)]))
    got = compute_MSM_batch(jobs)
    assert len(got) == 4
    for (b, s), g in zip(jobs, got):
        assert g == naive(b, s, A)
    assert compute_MSM_batch([]) == []
    # many accumulators (one per "proof"), one bad
    G = [U.get_random_point() for _ in range(12)]
    accs = []
    for k in range(5):
        a = [U.random_scalar() for _ in range(12)]
        acc = A.MSMAccumulator()
        C = naive(G, a, A)
        if k == 3:
            C = C + U.G1
        acc.accumulate_check(C, G, a)
        accs.append(acc)
    assert A.MSMAccumulator.verify_many(accs) == [True, True, True, False, True]


def test_batch_mul_patterns(api):
    """Vectorised `G1Point * Scalar` patterns of the callers against the host operators."""
    A, U = api
    from curdleproofs_pie_amd.msm_accumulator import batch_fold, batch_mul, batch_mul_same_scalar

    random.seed(8)
    n = 37
    L = [U.get_random_point() for _ in range(n)]
    Rr = [U.get_random_point() for _ in range(n)]
    L[5] = U.Z1
    Rr[7] = U.Z1
    sc = [U.random_scalar() for _ in range(n)]
    sc[3] = A.Scalar(0)
    gamma = U.random_scalar()
    from curdleproofs_pie_amd import _native as N

    ctx = N.default_context()
    for host_max in (0, -1):                       # 0: the GPU kernel; -1 (default): a call this small goes to the host's worker pool
        ctx.set_param("batch_mul_host_max", host_max)
        try:
            assert batch_mul(Rr, sc) == [r * s for r, s in zip(Rr, sc)]                       # grand_prod.py:64-71
            assert batch_mul_same_scalar(Rr, gamma) == [r * gamma for r in Rr]                # curdleproofs.py:310-311
            assert batch_fold(L, Rr, gamma) == [l + r * gamma for l, r in zip(L, Rr)]         # ipa.py:142-146
            assert batch_fold(L, L, A.Scalar(A.CURVE_ORDER - 1)) == [U.Z1] * n                # l + (-1) l == identity
            assert batch_fold(L, L, A.Scalar(1)) == [l + l for l in L]                        # doubling through the add
            assert batch_mul([], []) == []
        finally:
            ctx.set_param("batch_mul_host_max", -1)


def test_batch_to_compressed(api):
    """GPU batched compression == to_compressed_bytes, incl. identity and both sign flags; round trip through the GPU decoder."""
    A, U = api
    from curdleproofs_pie_amd.msm_accumulator import batch_from_compressed, batch_to_compressed

    random.seed(10)
    pts = [U.get_random_point() for _ in range(300)] + [U.Z1, U.G1, -U.G1]
    want = [bytes(p.to_compressed_bytes()) for p in pts]
    got = batch_to_compressed(pts)
    assert got == want
    assert batch_from_compressed(got) == pts
    assert batch_to_compressed([]) == []


def test_batch_from_compressed(api):
    """GPU batched decompression == the host decoder, incl. identity, both sign branches and rejections."""
    A, U = api
    from curdleproofs_pie_amd.msm_accumulator import batch_from_compressed

    random.seed(9)
    pts = [U.get_random_point() for _ in range(70)] + [U.Z1, U.G1, -U.G1]
    enc = [bytes(p.to_compressed_bytes()) for p in pts]
    assert sum(1 for e in enc if e[0] & 0x20) > 10 and sum(1 for e in enc if not e[0] & 0x20) > 10
    assert batch_from_compressed(enc) == pts
    assert batch_from_compressed(enc, checked=True) == pts
    assert batch_from_compressed([]) == []
    for bad in (bytes(48), bytes([0x9F]) + b"\\xff" * 47):
        with pytest.raises(ValueError):
            batch_from_compressed(enc[:3] + [bad] + enc[3:6])
    x = 1
    while O.fp_sqrt((x ** 3 + 4) % O.P) is not None:
        x += 1
    e = bytearray(x.to_bytes(48, "big")); e[0] |= 0x80
    with pytest.raises(ValueError):
        batch_from_compressed([bytes(e)])
    x = 1
    while True:   # on the curve, outside the subgroup: only the checked decoder rejects it
        y = O.fp_sqrt((x ** 3 + 4) % O.P)
        if y is not None and not O.g1_in_subgroup((x, y)):
            break
        x += 1
    e = bytearray(x.to_bytes(48, "big")); e[0] |= 0x80
    assert batch_from_compressed([bytes(e)]) == [A.G1Point.from_compressed_bytes_unchecked(bytes(e))]
    with pytest.raises(ValueError):
        batch_from_compressed(enc[:2] + [bytes(e)], checked=True)


def test_batch_sum_is_the_crs_point_sum(api):
    """crs.py:64-65: G_sum = reduce(a + b, vec_G, Z1), H_sum likewise -- against the oracle's group law, including groups
    that hit every exceptional case of the addition (equal points, opposite points, identities, an empty group)."""
    A, U = api
    from curdleproofs_pie_amd.msm_accumulator import batch_sum

    random.seed(11)
    vec_G = [U.get_random_point() for _ in range(124)]          # Whisk: ell = 124, n_blinders = 4
    vec_H = [U.get_random_point() for _ in range(4)]
    P = vec_G[0]
    groups = [vec_G, vec_H, [], [P], [P, P], [P, -P], [U.Z1, P, U.Z1], [P] * 70 + [-P] * 69, vec_G[:65], [U.Z1] * 3]
    got = batch_sum(groups)
    assert len(got) == len(groups)
    for g, r in zip(groups, got):
        want = None
        for p in g:
            want = O.g1_add(want, O.g1_decompress(bytes(p.to_compressed_bytes())))
        assert bytes(r.to_compressed_bytes()) == O.g1_compress(want)
    assert got[7] == P and got[5] == U.Z1 and got[2] == U.Z1
    assert batch_sum([]) == []


def test_msm_over_a_device_list(native_lib):
    """cg1_msm_multi_device (SURVEY 8(b) "multi-GPU variants taking a device list"): one MSM whose point shards live on several
    contexts -- here both on this GPU -- equals the single-context MSM and the oracle."""
    import ctypes

    from oracle import c_oracle as C

    N = native_lib
    c0, c1 = N.Context(0), N.Context(0)
    rng = random.Random(12)
    n0, n1 = 3000, 1777
    base = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(16)]
    raw = lambda p: p[0].to_bytes(48, "little") + p[1].to_bytes(48, "little")
    p96 = b"".join(raw(base[rng.randrange(16)]) for _ in range(n0 + n1))
    s32 = b"".join(rng.randint(0, O.R - 1).to_bytes(32, "little") for _ in range(n0 + n1))
    bufs = []
    for cx, lo, hi in ((c0, 0, n0), (c1, n0, n0 + n1)):
        dp, ds = cx.alloc(96 * (hi - lo)), cx.alloc(32 * (hi - lo))
        dp.upload(p96[96 * lo: 96 * hi]); ds.upload(s32[32 * lo: 32 * hi])
        bufs.append((dp, ds))
    blob = N.msm_multi_device([c0, c1], [b[0] for b in bufs], [b[1] for b in bufs], [n0, n1])
    out = ctypes.create_string_buffer(48)
    N.cg1_compress(out, blob)
    assert out.raw == C.compress(C.msm_bucket(p96, s32, n0 + n1))
    assert N.cg1_eq(blob, c0.msm_host(p96, s32, n0 + n1))
    with pytest.raises(N.NativeError):                       # one context twice: a context takes one call at a time
        N.msm_multi_device([c0, c0], [bufs[0][0]] * 2, [bufs[0][1]] * 2, [n0, n0])
    # an empty shard contributes the identity
    assert N.cg1_eq(N.msm_multi_device([c0, c1], [bufs[0][0], bufs[1][0]], [bufs[0][1], bufs[1][1]], [n0, 0]), c0.msm_host(p96, s32, n0))


def test_compute_msm_python_face_at_config2_size(api):
    """BASELINE config 2 through the reference's own signature: compute_MSM(bases, scalars) over 2^16 G1Point / Scalar OBJECTS
    (msm_accumulator.py:6-12; the objects are marshalled by one batched normalisation).  The bases are k_i * G, so the result
    must be exactly (sum k_i s_i) * G; a slice of the inputs is also summed by the oracle's naive loop."""
    A, U = api
    from curdleproofs_pie_amd.msm_accumulator import batch_mul
    from oracle import c_oracle as C

    rng = random.Random(16)
    n = 1 << 16
    ks = [A.Scalar(rng.randint(1, O.R - 1)) for _ in range(n)]
    bases = batch_mul([A.G1Point()] * n, ks)                 # get_random_point() = G * random_scalar() (util.py:67-68), vectorised
    scalars = [A.Scalar(rng.randint(0, O.R - 1)) for _ in range(n)]
    got = A.compute_MSM(bases, scalars)
    tot = sum(int(k) * int(s) for k, s in zip(ks, scalars)) % O.R
    assert bytes(got.to_compressed_bytes()) == O.g1_compress(O.g1_mul(O.G1_GEN, tot))
    # generators allowed, truncation to the shorter (msm_accumulator.py:10)
    assert A.compute_MSM(iter(bases), (s for s in scalars[: n - 5])) == got - A.compute_MSM(bases[n - 5:], scalars[n - 5:])
    m = 300
    from curdleproofs_pie_amd.py_arkworks_bls12381 import points_to_affine96
    want = C.compress(C.compute_msm(points_to_affine96(bases[:m]), b"".join(int(s).to_bytes(32, "little") for s in scalars[:m]), m))
    assert bytes(A.compute_MSM(bases[:m], scalars[:m]).to_compressed_bytes()) == want


def test_compute_msm_paths_blobs_and_resident_vectors(api):
    """compute_MSM's three ways in: a small call over host-normalised points (normal forms cached per object), a large one over
    the objects' own blobs (normalised on the device, k_prepare_blobs), and a base list that has become resident on the device
    (second sighting of the same objects).  Every path gives the reference loop's result."""
    A, U = api
    import curdleproofs_pie_amd.msm_accumulator as M
    import curdleproofs_pie_amd.py_arkworks_bls12381 as B

    want_of = lambda b, s: O.g1_compress(O.compute_MSM_fast([O.g1_decompress(bytes(x.to_compressed_bytes())) for x in b], [int(v) for v in s]))
    if B.lazy_enabled():
        # calls of the protocol's sizes are deferred values (evaluated by the GPU's batched MSM when their bytes are asked for); the
        # three immediate ways in serve calls above M.LAZY_MSM_MAX terms (and everything when deferral is off: the other parameter)
        deferred = True
    else:
        deferred = False
    for n, first_path in ((200, "affine"), (1500, "blobs")):
        M.clear_vec_cache()
        random.seed(21 + n)
        bases = [U.get_random_point() for _ in range(n)]                         # G * s: projective blobs, Z != 1
        bases[17] = U.Z1
        bases[40] = bases[3]
        seen = []
        for call in range(4):
            scalars = [U.random_scalar() for _ in range(n)]
            got = A.compute_MSM(bases, scalars)
            seen.append(M.last_path)
            assert bytes(got.to_compressed_bytes()) == want_of(bases, scalars)
        assert seen == (["deferred"] * 4 if deferred else [first_path, "resident", "resident", "resident"])
        # a prefix of the resident list is another sequence of objects: its own entry
        scalars = [U.random_scalar() for _ in range(n - 9)]
        assert bytes(A.compute_MSM(bases[: n - 9], scalars).to_compressed_bytes()) == want_of(bases[: n - 9], scalars)
        assert M.last_path == ("deferred" if deferred else first_path)
        # replacing an element of the caller's list must not hit the stale resident vector
        bases[5] = U.get_random_point()
        scalars = [U.random_scalar() for _ in range(n)]
        assert bytes(A.compute_MSM(bases, scalars).to_compressed_bytes()) == want_of(bases, scalars)
        assert M.last_path == ("deferred" if deferred else first_path)
        if n > 1024 and not deferred:
            # normal forms (decoded points) skip the device inversion
            dec = [A.G1Point.from_compressed_bytes_unchecked(b.to_compressed_bytes()) for b in bases]
            assert bytes(A.compute_MSM(dec, scalars).to_compressed_bytes()) == want_of(bases, scalars)
            assert M.last_path == "blobs_normalised"
            assert A.compute_MSM(dec, scalars) == A.compute_MSM(bases, scalars) and M.last_path == "resident"
    # every size 1 .. 40 and a few around the slice edges of the small kernel
    pool = [U.get_random_point() for _ in range(64)] + [U.Z1]
    for m in list(range(1, 41)) + [127, 128, 129, 255, 256, 257, 1023, 1024, 1025]:
        b = [pool[random.randrange(len(pool))] for _ in range(m)]
        s = [U.random_scalar() for _ in range(m)]
        assert bytes(A.compute_MSM(b, s).to_compressed_bytes()) == want_of(b, s), m
    M.clear_vec_cache()


def test_msm_blobs_many_points_per_lane(native_lib):
    """cg1_msm_blobs above 2^17 points: k_prepare_blobs inverts once per lane over K > 1 consecutive points (identities inside the
    runs, a ragged last lane).  Bases are tiled from 61 projective multiples k_j G, so the result is (sum s_i k_(i mod 61)) G."""
    import ctypes

    N = native_lib
    ctx = N.Context(0)
    rng = random.Random(22)
    ks = [rng.randint(1, O.R - 1) for _ in range(61)]
    ks[13] = 0                                                            # the identity (Z = 0) inside every run
    g = ctypes.create_string_buffer(144)
    N.cg1_generator(g)
    blobs = []
    for k in ks:
        out = ctypes.create_string_buffer(144)
        N.cg1_mul(out, g.raw, k.to_bytes(32, "little"))
        blobs.append(out.raw)
    assert blobs[0][96:] != blobs[1][96:]                                 # genuinely projective
    n = (1 << 17) * 3 + 77                                                # K = 4, last lane ragged
    raw = b"".join(blobs) * (n // 61 + 1)
    raw = raw[: 144 * n]
    sc = [rng.randint(0, O.R - 1) for _ in range(n)]
    s32 = b"".join(v.to_bytes(32, "little") for v in sc)
    got = ctx.msm_blobs(raw, s32, n, False)
    tot = sum(s * ks[i % 61] for i, s in enumerate(sc)) % O.R
    out = ctypes.create_string_buffer(48)
    N.cg1_compress(out, got)
    assert out.raw == O.g1_compress(O.g1_mul(O.G1_GEN, tot))
    # the resident form of the same blobs, and a window of it
    vec = ctx.vec(raw, n, False)
    assert N.cg1_eq(ctx.msm_vec(vec, s32, n), got) == 1
    first, m = 12345, 5000
    part = ctx.msm_vec(vec, s32[32 * first: 32 * (first + m)], m, first)
    tot = sum(sc[i] * ks[i % 61] for i in range(first, first + m)) % O.R
    N.cg1_compress(out, part)
    assert out.raw == O.g1_compress(O.g1_mul(O.G1_GEN, tot))
    with pytest.raises(N.NativeError):
        ctx.msm_vec(vec, s32, n, 1)                                       # window past the end
    vec.free()


def test_drop_in_from_four_threads(api):
    """The wheel's values may be used from any thread; so may this backend: 4 Python threads, 1 000 iterations between them, each a
    compute_MSM of a mixed size -- deferred (k_msm_small through the batched evaluation), immediate over host-normalised points, over
    blobs, over a base list resident on the device -- or a whole MSMAccumulator sequence, every result checked against the oracle's
    closed form (the bases are k_i G, so the sum is (sum k_i s_i) G).  One lock serialises the device calls (msm_accumulator._LOCK)."""
    import threading

    A, U = api
    import curdleproofs_pie_amd.msm_accumulator as M
    from curdleproofs_pie_amd.msm_accumulator import batch_mul

    rng = random.Random(31)
    n_max = 3000
    ks = [rng.randint(1, O.R - 1) for _ in range(n_max)]
    pts = batch_mul([A.G1Point()] * n_max, [A.Scalar(k) for k in ks])
    shared = pts[:2500]                                       # one list object used by every thread: becomes resident on the device
    errors = []
    want_cache = {}
    lock = threading.Lock()

    def expect(tot):
        with lock:
            w = want_cache.get(tot)
        if w is None:
            w = O.g1_compress(O.g1_mul(O.G1_GEN, tot % O.R))
            with lock:
                want_cache[tot] = w
        return w

    def worker(t):
        r = random.Random(100 + t)
        try:
            for it in range(250):
                kind = r.choice(["small", "small", "mid", "blobs", "resident", "acc"])
                if kind == "acc":
                    acc = A.MSMAccumulator()
                    for _ in range(3):
                        m = r.choice([4, 17, 60])
                        idx = [r.randrange(n_max) for _ in range(m)]
                        sc = [r.randint(0, O.R - 1) for _ in range(m)]
                        C = A.compute_MSM([pts[i] for i in idx], [A.Scalar(s) for s in sc])
                        acc.accumulate_check(C, [pts[i] for i in idx], [A.Scalar(s) for s in sc])
                    acc.verify()
                    bad = A.MSMAccumulator()
                    bad.accumulate_check(pts[0], [pts[1]], [A.Scalar(1)])
                    try:
                        bad.verify()
                        errors.append((t, it, "a wrong check was accepted"))
                    except AssertionError:
                        pass
                    continue
                if kind == "resident":
                    bases, idx = shared, range(2500)
                else:
                    m = {"small": r.choice([1, 5, 37, 300]), "mid": r.choice([1100, 1900]), "blobs": r.choice([2100, 2900])}[kind]
                    lo = r.randrange(n_max - m + 1)
                    idx = range(lo, lo + m)
                    bases = pts[lo: lo + m]
                sc = [r.randint(0, O.R - 1) for _ in idx]
                got = A.compute_MSM(bases, [A.Scalar(s) for s in sc])
                if bytes(got.to_compressed_bytes()) != expect(sum(ks[i] * s for i, s in zip(idx, sc))):
                    errors.append((t, it, kind, len(sc)))
        except Exception as e:           # noqa: BLE001 -- reported below
            import traceback

            errors.append((t, repr(e), traceback.format_exc()[-600:]))

    M.clear_vec_cache()
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert errors == []
    M.clear_vec_cache()


def test_compute_msm_over_deferred_bases_outside_g1(api):
    """compute_MSM whose bases are themselves deferred values (a prover's folded `G_L[i] + G_R[i] * gamma`, ipa.py:142-146) over points
    decoded unchecked: over bases of G1 the coefficients fold into the scalars; a base carrying the order-3 point (0, 2) must be evaluated
    first (`(P * a) * b != P * (a b mod r)` there) -- either way the oracle's point comes back."""
    A, U = api
    rng = random.Random(51)
    T3 = (0, 2)
    raw = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(12)]
    raw[3] = O.g1_add(raw[3], T3)
    raw[8] = O.g1_add(raw[8], O.g1_neg(T3))
    pts = [A.G1Point.from_compressed_bytes_unchecked(O.g1_compress(p)) for p in raw]
    gamma = rng.randrange(2, O.R)
    half = 6
    folded = [pts[i] + pts[half + i] * A.Scalar(gamma) for i in range(half)]                  # deferred (or computed at once: the other parameter)
    want_folded = [O.g1_add(raw[i], O.g1_mul(raw[half + i], gamma)) for i in range(half)]
    sc = [rng.randrange(O.R) for _ in range(half)]
    got = A.compute_MSM(folded, [A.Scalar(s) for s in sc])
    want = None
    for p, s in zip(want_folded, sc):
        want = O.g1_add(want, O.g1_mul(p, s))
    assert bytes(got.to_compressed_bytes()) == O.g1_compress(want)
    # a second round of folding over the first (depth 2), then an MSM over it
    g2 = rng.randrange(2, O.R)
    folded2 = [folded[i] + folded[3 + i] * A.Scalar(g2) for i in range(3)]
    want2 = None
    sc2 = [rng.randrange(O.R) for _ in range(3)]
    for i in range(3):
        f = O.g1_add(want_folded[i], O.g1_mul(want_folded[3 + i], g2))
        want2 = O.g1_add(want2, O.g1_mul(f, sc2[i]))
    assert bytes(A.compute_MSM(folded2, [A.Scalar(s) for s in sc2]).to_compressed_bytes()) == O.g1_compress(want2)


def test_accumulator_with_points_outside_g1(api):
    """accumulate_check over bases decoded unchecked that carry the order-3 point (0, 2): the reference computes A_c += C * rho and
    compares it with the MSM over the merged (mod r) scalars (msm_accumulator.py:45,56-68) as curve points, whatever their order -- so
    `-(rho * C)` must enter the final MSM as rho * (-C), not as (r - rho) * C.  Honest and dishonest checks, leaf and deferred left-hand
    sides, against the oracle's restatement of the class (found by tools/gpu_lazy_fuzz.py)."""
    A, U = api
    import curdleproofs_pie_amd.msm_accumulator as M
    from oracle import py_arkworks_shim as S

    rng = random.Random(61)
    T3 = (0, 2)
    raw = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(6)]
    raw[1] = O.g1_add(raw[1], T3)
    raw[4] = O.g1_add(raw[4], O.g1_neg(T3))
    enc = [O.g1_compress(p) for p in raw]

    def omsm(bs, ss):
        cur = S.G1Point.identity()
        for b, s in zip(bs, ss):
            cur = cur + b * s
        return cur

    for case in range(12):
        mine = [A.G1Point.from_compressed_bytes_unchecked(e) for e in enc]
        ref = [S.G1Point.from_compressed_bytes_unchecked(e) for e in enc]
        acc = A.MSMAccumulator()
        o_A, o_map = S.G1Point.identity(), {}
        honest = case % 3 != 2
        for call in range(2):
            idx = [rng.randrange(6) for _ in range(rng.choice([1, 3, 5]))]
            sc = [rng.randrange(O.R) for _ in idx]
            C_o = omsm([ref[t] for t in idx], [S.Scalar(s) for s in sc])
            if case % 2:                                       # a deferred left-hand side (compute_MSM of the same terms) ...
                C_m = A.compute_MSM([mine[t] for t in idx], [A.Scalar(s) for s in sc])
            else:                                              # ... or a decoded leaf
                C_m = A.G1Point.from_compressed_bytes_unchecked(bytes(C_o.to_compressed_bytes()))
            if not honest and call == 1:
                C_o = C_o + ref[0]; C_m = C_m + mine[0]
            rho = rng.randrange(1, O.R)
            orig = M.random_scalar
            M.random_scalar = lambda rho=rho: A.Scalar(rho)
            try:
                acc.accumulate_check(C_m, [mine[t] for t in idx], [A.Scalar(s) for s in sc])
            finally:
                M.random_scalar = orig
            o_A = o_A + C_o * S.Scalar(rho)
            for t, s in zip(idx, sc):
                k = bytes(ref[t].to_compressed_bytes())
                o_map[k] = o_map.get(k, S.Scalar(0)) + S.Scalar(rho) * S.Scalar(s)
        keys, vals = zip(*o_map.items())
        want = omsm([S.G1Point.from_compressed_bytes_unchecked(k) for k in keys], vals) == o_A
        try:
            acc.verify(); got = True
        except AssertionError:
            got = False
        assert got == want, (case, got, want)
