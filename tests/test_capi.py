"""The C-ABI library loads and exports every symbol include/curdle_g1.h declares (no compute calls that
need a GPU).  Also: the product path fails loudly, with no CPU fallback, when no GPU is visible."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "curdle_g1.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cg1_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(native_lib):
    names = declared_symbols()
    assert len(names) >= 30
    lib = ctypes.CDLL(native_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/curdle_g1.h but not exported"
    assert sorted(native_lib.EXPORTED_SYMBOLS) == names


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "curdleproofs_pie_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("test_oracle", ""), f"{f} mentions the oracle"


def test_oracle_does_not_import_product():
    """The checker stands on its own: no file under oracle/ imports (or names) the product package, so no oracle verdict
    can be produced by the code under test."""
    odir = os.path.join(ROOT, "oracle")
    for f in os.listdir(odir):
        if f.endswith((".py", ".c", ".h")) or f == "Makefile":
            text = open(os.path.join(odir, f)).read()
            assert "import curdleproofs_pie_amd" not in text and "from curdleproofs_pie_amd" not in text and "libcurdle_g1" not in text, f


def test_msm_fails_loudly_without_gpu(native_lib):
    if native_lib.cg1_device_count() > 0:
        pytest.skip("a GPU is visible; the no-GPU failure mode is checked on the CPU box")
    from curdleproofs_pie_amd import G1Point, Scalar, compute_MSM, MSMAccumulator

    with pytest.raises(native_lib.NativeError):
        compute_MSM([G1Point()], [Scalar(3)])
    acc = MSMAccumulator()
    acc.accumulate_check(G1Point() * Scalar(3), [G1Point()], [Scalar(3)])
    with pytest.raises(native_lib.NativeError):
        acc.verify()
    assert compute_MSM([], []) == G1Point.identity()      # msm_accumulator.py:9 -- no device work for n = 0
