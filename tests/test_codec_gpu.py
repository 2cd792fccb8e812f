"""GPU wire codec (k_batch_decompress / k_batch_compress) through the C ABI against the CPU oracle's codec
(oracle/bls12_381.py g1_decompress / g1_compress == util.py:27-36 -> G1Point.to/from_compressed_bytes[_unchecked]).
Inputs are oracle-made points and hand-made bad encodings: no product host arithmetic on the expected side."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bls12_381 as O  # noqa: E402
from oracle import c_oracle as C  # noqa: E402

pytestmark = pytest.mark.gpu


def raw96(pt):
    return bytes(96) if pt is None else pt[0].to_bytes(48, "little") + pt[1].to_bytes(48, "little")


@pytest.fixture(scope="module")
def ctx(native_lib):
    return native_lib.Context(0)


@pytest.fixture(scope="module")
def points():
    rng = random.Random(77)
    pts = [O.g1_mul(O.G1_GEN, rng.randint(1, O.R - 1)) for _ in range(96)]
    pts += [O.g1_neg(p) for p in pts[:32]] + [None, O.G1_GEN, O.g1_neg(O.G1_GEN), None]
    return pts


def gpu_decompress(N, ctx, encs, check):
    """-> (affine96 list, status list) straight from cg1_batch_decompress_device (per-point status, nothing raised)."""
    n = len(encs)
    d_in, d_out, d_st = ctx.alloc(48 * n), ctx.alloc(96 * n), ctx.alloc(n)
    d_in.upload(b"".join(encs))
    ctx.check(N.cg1_batch_decompress_device(ctx.handle, d_in.ptr, d_out.ptr, d_st.ptr, n, 1 if check else 0))
    out, st = d_out.download(96 * n), d_st.download(n)
    return [out[96 * i: 96 * i + 96] for i in range(n)], list(st)


def off_subgroup_point():
    x = 1
    while True:
        y = O.fp_sqrt((x ** 3 + 4) % O.P)
        if y is not None and not O.g1_in_subgroup((x, y)):
            return (x, y)
        x += 1


@pytest.mark.parametrize("check", [False, True])
def test_gpu_decompress_equals_oracle(native_lib, ctx, points, check):
    N = native_lib
    encs = [O.g1_compress(p) for p in points]
    assert sum(1 for e in encs if e[0] & 0x20) > 20 and sum(1 for e in encs if not e[0] & 0x20) > 20
    got, st = gpu_decompress(N, ctx, encs, check)
    assert st == [0] * len(encs)
    assert got == [raw96(O.g1_decompress(e, check)) for e in encs] == [raw96(p) for p in points]


def test_gpu_decompress_rejections_equal_oracle(native_lib, ctx, points):
    """Every malformed / off-curve / off-subgroup encoding gets the verdict the oracle decoder gives, per point, in a
    batch that also holds valid points (one bad lane must not disturb its neighbours)."""
    N = native_lib
    x_off = 1
    while O.fp_sqrt((x_off ** 3 + 4) % O.P) is not None:
        x_off += 1
    e_off = bytearray(x_off.to_bytes(48, "big")); e_off[0] |= 0x80
    tors = off_subgroup_point()
    bad = [
        bytes(48),                                           # compression flag clear
        bytes([0xE0]) + bytes(47),                           # infinity + sign
        bytes([0xC0]) + bytes(46) + b"\x01",                 # infinity with non-zero x
        bytes([0x9F]) + b"\xff" * 47,                        # x >= p
        bytes([0x80 | (O.P >> 376)]) + (O.P % (1 << 376)).to_bytes(47, "big"),     # x == p
        bytes(e_off),                                        # x^3 + 4 is not a square
        O.g1_compress(tors), O.g1_compress(O.g1_neg(tors)),  # on the curve, outside G1
    ]
    encs = []
    for i, b in enumerate(bad):
        encs += [O.g1_compress(points[i]), b]
    for check in (False, True):
        got, st = gpu_decompress(N, ctx, encs, check)
        for e, g, s in zip(encs, got, st):
            try:
                want = raw96(O.g1_decompress(e, check))
                assert s == 0 and g == want, (e.hex(), check)
            except ValueError:
                assert s != 0 and g == bytes(96), (e.hex(), check)
            rc, aff = C.decompress(e, check)                 # the C restatement agrees with the Python one
            assert (rc == 0) == (s == 0) and aff == g


def test_gpu_compress_equals_oracle(native_lib, ctx, points):
    N = native_lib
    n = len(points)
    d_in, d_out = ctx.alloc(96 * n), ctx.alloc(48 * n)
    d_in.upload(b"".join(raw96(p) for p in points))
    ctx.check(N.cg1_batch_compress_device(ctx.handle, d_in.ptr, d_out.ptr, n))
    got = d_out.download(48 * n)
    assert [got[48 * i: 48 * i + 48] for i in range(n)] == [O.g1_compress(p) for p in points]


def test_python_face_codec_equals_oracle(native_lib, points):
    """batch_from_compressed / batch_to_compressed (the Python face) against the oracle's bytes."""
    from curdleproofs_pie_amd.msm_accumulator import batch_from_compressed, batch_to_compressed

    encs = [O.g1_compress(p) for p in points]
    objs = batch_from_compressed(encs)
    assert batch_to_compressed(objs) == encs
    assert [bytes(p.to_compressed_bytes()) for p in objs] == encs      # host encoder agrees too
    with pytest.raises(ValueError):
        batch_from_compressed(encs[:3] + [O.g1_compress(off_subgroup_point())], checked=True)
