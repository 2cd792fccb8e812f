import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")      # as an application would (N.tune_runtime()): before the first HIP call
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def native_lib():
    """libcurdle_g1.so, built on demand (hipcc cross-compiles without a GPU)."""
    from curdleproofs_pie_amd import build as B

    B.build(verbose=False)
    from curdleproofs_pie_amd import _native

    return _native


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "msm_vectors.json")) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="session")
def fp28_harness():
    """The device field/group headers compiled for the host with range checks (tests/native)."""
    import ctypes

    src = os.path.join(ROOT, "tests", "native", "fp28_harness.cpp")
    out_dir = os.path.join(ROOT, "tests", "native", "_build")
    out = os.path.join(out_dir, "libfp28_harness.so")
    deps = [src] + [os.path.join(ROOT, "curdleproofs_pie_amd", "csrc", f) for f in ("fp28.h", "g1_xyzz.h", "bls_consts.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-DCG1_CHECK_BOUNDS", "-shared", "-fPIC", src, "-o", out])
    return ctypes.CDLL(out)


def raw96(pt):
    return bytes(96) if pt is None else pt[0].to_bytes(48, "little") + pt[1].to_bytes(48, "little")


def from_raw96(b):
    if b == bytes(96):
        return None
    return (int.from_bytes(b[:48], "little"), int.from_bytes(b[48:96], "little"))
