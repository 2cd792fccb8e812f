"""curdleproofs_pie_amd -- MI355X-native BLS12-381 G1 / MSM engine behind curdleproofs.pie's
compute_MSM / MSMAccumulator and the G1Point / Scalar surface of py_arkworks_bls12381.

Host code is plain Python + ctypes over libcurdle_g1.so (hand-written HIP for gfx950 + host C++);
no torch, no Triton, no CPU fallback for the batched paths.

Attributes resolve lazily so that `python -m curdleproofs_pie_amd.build` can run before the shared
library exists; touching any of them without the library raises ImportError (never a fallback).
"""
__all__ = ["G1Point", "Scalar", "CURVE_ORDER", "MSMAccumulator", "compute_MSM", "compute_MSM_batch"]


def __getattr__(name):
    if name in ("G1Point", "Scalar", "CURVE_ORDER"):
        from . import py_arkworks_bls12381 as m

        return getattr(m, name)
    if name in ("MSMAccumulator", "compute_MSM", "compute_MSM_batch"):
        from . import msm_accumulator as m

        return getattr(m, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
