"""Backend selection for the golden-fixture generators (build container only: they import /root/reference).

The reference's arithmetic lives in the Rust wheel py_arkworks_bls12381 0.3.5, which cannot run here.  The generators
inject a stand-in as `sys.modules["py_arkworks_bls12381"]` before importing the reference:

    oracle   (default)  oracle/py_arkworks_shim.py -- pure-Python big integers over oracle/bls12_381.py; nothing in a
                        fixture then comes out of the product's arithmetic
    product             curdleproofs_pie_amd.py_arkworks_bls12381 (host C++) -- ~40x faster; used to show that both
                        backends produce byte-identical files (tests/test_golden_backends.py)

    python tests/golden/gen_shuffle_golden.py [--backend oracle|product] [--out PATH]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
for _p in ("/root/reference/curdleproofs", "/root/reference/merlin_transcripts"):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def _arg(flag, default):
    if flag in sys.argv:
        i = sys.argv.index(flag)
        return sys.argv[i + 1]
    return default


BACKEND = os.environ.get("GOLDEN_BACKEND") or _arg("--backend", "oracle")
OUT = _arg("--out", None)


def inject():
    """Install the chosen G1Point/Scalar module as `py_arkworks_bls12381`; returns its name for the fixture header."""
    if "py_arkworks_bls12381" in sys.modules:
        return sys.modules["py_arkworks_bls12381"].__name__
    if BACKEND == "oracle":
        import oracle.py_arkworks_shim as backend
    elif BACKEND == "product":
        import curdleproofs_pie_amd.py_arkworks_bls12381 as backend
    else:
        raise SystemExit(f"unknown backend {BACKEND!r}")
    sys.modules["py_arkworks_bls12381"] = backend
    return backend.__name__


def out_path(default_name):
    return OUT or os.path.join(os.path.dirname(os.path.abspath(__file__)), default_name)
