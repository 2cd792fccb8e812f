"""The gfx950 field / group code (csrc/fp28.h, csrc/g1_xyzz.h) compiled for the host with CG1_CHECK_BOUNDS:
every 64-bit column accumulator, lazy add and lazy subtract is range-checked (abort on violation) and the
results are compared with the big-int oracle.  This is how the device arithmetic is validated without a GPU."""
import ctypes
import random

from conftest import raw96
from oracle import bls12_381 as O

P = O.P


def b48(v):
    return v.to_bytes(48, "little")


def parse_xyzz(buf):
    b = buf.raw
    if int.from_bytes(b[192:196], "little"):
        return None
    # exported coordinates are canonical and in the host library's Montgomery form (x * 2^384 mod p)
    X, Y, ZZ, ZZZ = (int.from_bytes(b[48 * i: 48 * i + 48], "little") for i in range(4))
    assert max(X, Y, ZZ, ZZZ) < P
    rinv = pow(1 << 384, -1, P)
    X, Y, ZZ, ZZZ = (v * rinv % P for v in (X, Y, ZZ, ZZZ))
    assert (pow(ZZ, 3, P) - pow(ZZZ, 2, P)) % P == 0
    return (X * pow(ZZ, -1, P) % P, Y * pow(ZZZ, -1, P) % P)


def test_field_ops(fp28_harness):
    L = fp28_harness
    rng = random.Random(1)
    o = ctypes.create_string_buffer(48)
    edge = [0, 1, 2, P - 1, P - 2, (1 << 380), (1 << 381) - 1 - ((1 << 381) - 1 >= P) * 0 if ((1 << 381) - 1) < P else P - 3]
    vals = edge + [rng.randrange(P) for _ in range(300)]
    for i, a in enumerate(vals):
        b = vals[(7 * i + 3) % len(vals)]
        c = vals[(11 * i + 5) % len(vals)]
        L.t_fp_mul(b48(a), b48(b), o); assert int.from_bytes(o.raw, "little") == a * b % P
        L.t_fp_sqr(b48(a), o); assert int.from_bytes(o.raw, "little") == a * a % P
        L.t_fp_submul(b48(a), b48(b), b48(c), o); assert int.from_bytes(o.raw, "little") == (a - b) * c % P
        L.t_fp_addmul(b48(a), b48(b), b48(c), o); assert int.from_bytes(o.raw, "little") == (a + b) * c % P
        assert L.t_fp_is_zero_diff(b48(a), b48(b)) == (1 if a == b else 0)
        assert L.t_fp_is_zero_diff(b48(a), b48(a)) == 1
    for a in [1, 2, P - 1] + [rng.randrange(1, P) for _ in range(10)]:
        L.t_fp_inv(b48(a), o); assert int.from_bytes(o.raw, "little") == pow(a, -1, P)


def test_worst_case_limb_magnitudes_do_not_overflow(fp28_harness):
    # aborts the process (CG1_ASSERT) if any column accumulator would exceed 64 bits at the admitted maxima
    assert fp28_harness.t_worst_case_bounds() & 2


def test_group_law_with_exceptional_cases(fp28_harness):
    L = fp28_harness
    rng = random.Random(2)
    pts = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(24)]
    o = ctypes.create_string_buffer(196)

    def check(seq, negs):
        buf = b"".join(raw96(p) for p in seq)
        want = None
        for p, s in zip(seq, negs):
            want = O.g1_add(want, O.g1_neg(p) if s else p)
        L.t_madd_seq(buf, bytes(negs), len(seq), o); assert parse_xyzz(o) == want
        L.t_chunk_seq(buf, bytes(negs), len(seq), o); assert parse_xyzz(o) == want      # first pair affine+affine, as k_accumulate
        L.t_add_tree(buf, bytes(negs), len(seq), o); assert parse_xyzz(o) == want

    check(pts, [rng.randrange(2) for _ in pts])
    check([], [])
    check([pts[0]], [1])
    check([pts[0], pts[0]], [0, 0])                       # P + P through the mixed add
    check([pts[0], pts[0]], [0, 1])                       # P + (-P)
    check([pts[0], pts[0]], [1, 1])                       # (-P) + (-P): both lazily negated
    check([pts[0], pts[0], pts[3]], [1, 0, 1])            # identity from the first pair, then continue
    check([pts[0], pts[1]], [1, 0])
    check([pts[0]] * 9, [0] * 9)
    check([pts[0], pts[0], pts[0], pts[1], pts[0]], [0, 1, 0, 0, 0])
    s01 = O.g1_add(pts[0], pts[1])
    check([pts[0], pts[1], s01], [0, 0, 0])               # accumulator == next point -> doubling branch
    check([pts[0], pts[1], s01], [0, 0, 1])               # accumulator == -next point -> identity
    check([pts[0], pts[1], s01, pts[5]], [0, 0, 1, 0])    # continue after hitting the identity
    for k in [0, 1, 2, 99, O.R - 1, rng.randrange(O.R)]:
        L.t_scalar_mul(raw96(O.G1_GEN), k.to_bytes(32, "little"), o)
        assert parse_xyzz(o) == O.g1_mul(O.G1_GEN, k)
    seq = pts[:13]
    L.t_running_sum(b"".join(raw96(p) for p in seq), len(seq), o)
    want = None
    for i, p in enumerate(seq):
        want = O.g1_add(want, O.g1_mul(p, i + 1))
    assert parse_xyzz(o) == want


def test_jacobian_ops_and_subgroup_test(fp28_harness):
    """jacp_dbl / jacp_madd / jacp_add (the subgroup test's arithmetic) against the oracle's group law, and g1_in_subgroup against
    r * P == identity on points of G1, random curve points, points of order 3 / 11 / 10177 and sums of those with points of G1."""
    L = fp28_harness
    rng = random.Random(9)
    o = ctypes.create_string_buffer(196)
    t3 = (0, 2)                                                          # order 3 (tests/golden/torsion_vectors.json "t3")
    assert O.g1_is_on_curve(t3) and O.g1_add(O.g1_add(t3, t3), t3) is None

    def curve_point():
        while True:
            x = rng.randrange(P)
            y = O.fp_sqrt((x * x * x + 4) % P)
            if y is not None:
                return (x, y if rng.randrange(2) else P - y)

    def mul_any(pt, k):                                                  # k * pt WITHOUT reducing k mod r (pt may lie outside G1)
        acc, base = O.JAC_INF, O.jac_from_affine(pt)
        for bit in bin(k)[2:] if k else "":
            acc = O.jac_double(acc)
            if bit == "1":
                acc = O.jac_add(acc, base)
        return O.jac_to_affine(acc)

    g1 = [O.g1_mul(O.G1_GEN, rng.randrange(1, O.R)) for _ in range(6)]
    wild = [curve_point() for _ in range(6)]
    for base in (O.G1_GEN, g1[0], wild[0], t3):
        for k in [0, 1, 2, 3, 4, 5, 6, 7, 99, O.R - 1, O.R, O.R + 1, (1 << 255) - 1, rng.randrange(1 << 255)]:
            L.t_scalar_mul_jac(raw96(base), k.to_bytes(32, "little"), o)
            assert parse_xyzz(o) == mul_any(base, k), (base == t3, k)

    def tree(seq, negs):
        want = None
        for p, s in zip(seq, negs):
            want = O.g1_add(want, O.g1_mul(O.g1_neg(p) if s else p, 2))
        L.t_add_tree_jac(b"".join(raw96(p) for p in seq), bytes(negs), len(seq), o)
        assert parse_xyzz(o) == want

    tree(g1 + wild, [rng.randrange(2) for _ in range(12)])
    tree([], [])
    tree([g1[0]], [1])
    tree([g1[0], g1[0]], [0, 0])                                         # doubling branch of jacp_add
    tree([g1[0], g1[0]], [0, 1])                                         # cancellation
    tree([g1[0], g1[0], g1[1]], [0, 1, 0])                               # identity operand
    tree([t3, t3], [0, 0])
    tree([t3, t3, t3], [0, 0, 0])
    tree([g1[0], g1[1], g1[0], g1[1]], [0, 0, 1, 1])

    # the cofactor's prime factors: (z - 1)^2 = 3 h, z - 1 = -(3 * 11 * 10177 * 859267 * 52437899)
    zm1 = 0xd201000000010001
    assert zm1 == 3 * 11 * 10177 * 859267 * 52437899
    h = zm1 * zm1 // 3
    small = [t3]
    for ell in (11, 10177):
        while True:
            t = mul_any(curve_point(), O.R * (h // (ell * ell)))          # the cofactor group is Z/((z-1)/3) x Z/(z-1): exponent ell, order ell^2
            if t is not None:
                break
        assert mul_any(t, ell) is None
        small.append(t)
    cases = g1 + wild + small + [O.g1_add(t, g) for t, g in zip(small, g1)] + [O.g1_add(small[0], small[1]), mul_any(wild[1], O.R), mul_any(wild[2], h)]
    for pt in cases:
        if pt is None:
            continue
        assert O.g1_is_on_curve(pt)
        assert bool(L.t_in_subgroup(raw96(pt))) == O.g1_in_subgroup(pt), pt
    assert [bool(L.t_in_subgroup(raw96(p))) for p in g1] == [True] * 6 and not any(L.t_in_subgroup(raw96(p)) for p in wild + small)
