"""The reference's own backend-call sequence for one ell = 124 Whisk shuffle proof (tests/golden/call_trace_ell124.*, recorded by
tests/golden/gen_call_trace.py from the unmodified reference over the oracle backend) replayed through the product:

  CPU  (no GPU): the G1Point operators on the host library, with compute_MSM / MSMAccumulator restated naively over those operators
        (msm_accumulator.py:6-12, :32-68) -- every recorded output (1 184 + 847 compressions, equality results, the verdict) comes back;
  GPU  the product's own compute_MSM / MSMAccumulator (Python face -> C ABI -> HIP kernels) in their place: the same outputs,
        bit for bit -- the drop-in on the reference's unchanged control flow.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def trace(native_lib):
    import replay_call_trace as RT

    doc, blob = RT.load()
    assert doc["backend"] == "oracle.py_arkworks_shim"          # no expected byte comes out of product arithmetic
    assert doc["verdict"] is True and doc["ell"] == 124
    assert doc["counts"]["verify"]["acc_check"] == 8 and doc["counts"]["verify"]["acc_verify"] == 1
    assert doc["counts"]["prove"]["msm"] == 88 and doc["counts"]["verify"]["msm"] == 10
    return RT, doc, blob


def test_trace_replays_on_the_host_face(trace):
    RT, doc, blob = trace
    import curdleproofs_pie_amd.py_arkworks_bls12381 as B

    def naive_msm(bases, scalars):                               # msm_accumulator.py:6-12
        cur = B.G1Point.identity()
        for b, s in zip(bases, scalars):
            cur = cur + b * s
        return cur

    box = {}

    class NaiveAccumulator:                                      # msm_accumulator.py:32-68
        def __init__(self):
            self.A_c = B.G1Point.identity()
            self.map = {}

        def accumulate_check(self, C, bases, scalars):
            rho = box.pop("rho")
            self.A_c = self.A_c + C * rho
            for b, s in zip(bases, scalars):
                if b == B.G1Point.identity():
                    continue
                k = bytes(b.to_compressed_bytes())
                self.map[k] = self.map.get(k, B.Scalar(0)) + rho * s

        def verify(self):
            keys, vals = zip(*self.map.items())
            assert naive_msm([B.G1Point.from_compressed_bytes_unchecked(k) for k in keys], vals) == self.A_c

    rp = RT.Replayer(doc, blob, B.G1Point, B.Scalar, naive_msm, NaiveAccumulator, lambda s: box.__setitem__("rho", s))
    for ph in ("setup", "prove", "verify"):
        rp.prepare(doc[ph])
        rp.run(doc[ph])
    assert rp.mismatches == []
    assert len(rp.vals) == sum(1 for ph in ("setup", "prove", "verify") for o in doc[ph] if o[0] in ("gen", "id", "dec", "add", "sub", "neg", "mul", "msm"))


@pytest.mark.gpu
@pytest.mark.parametrize("lazy", [True, False], ids=["deferred", "eager"])
def test_trace_replays_through_the_gpu_backend(trace, lazy):
    RT, doc, blob = trace
    import curdleproofs_pie_amd.msm_accumulator as M
    import curdleproofs_pie_amd.py_arkworks_bls12381 as B

    prev = B.set_lazy(lazy)
    try:
        _replay_on_gpu(RT, doc, blob, M, B, lazy)
    finally:
        B.set_lazy(prev)


def _replay_on_gpu(RT, doc, blob, M, B, lazy):
    M.clear_vec_cache()
    paths = {}
    before = dict(B.stats)
    for rep in range(2):                                         # twice: the second time CRS vectors are resident on the device
        rp = RT.product_replayer(doc, blob)
        orig = rp.compute_MSM

        def counted(bases, scalars, orig=orig):
            r = orig(bases, scalars)
            paths[M.last_path] = paths.get(M.last_path, 0) + 1
            return r

        rp.compute_MSM = counted
        for ph in ("setup", "prove", "verify"):
            rp.prepare(doc[ph])
            rp.run(doc[ph])
        assert rp.mismatches == [], rp.mismatches[:3]
    if lazy:
        # every compute_MSM of the protocol's sizes was deferred and evaluated on the GPU in groups (the L / R points of a halving round
        # together); the 585 + 248 + 133 decodings were validated one by one and their square roots taken in a few batches
        assert set(paths) == {"deferred"} and B.stats["flush_device"] > before["flush_device"]
        assert B.stats["flushed_values"] - before["flushed_values"] > 4 * (B.stats["flushes"] - before["flushes"])
        assert B.stats["decoded"] - before["decoded"] >= 2 * 700 and B.stats["decode_batches"] - before["decode_batches"] < 60
    else:
        assert paths.get("resident", 0) > 0 and paths.get("affine", 0) > 0       # both ways into the small-MSM kernel were taken
    M.clear_vec_cache()
