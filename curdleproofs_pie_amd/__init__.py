"""curdleproofs_pie_amd -- MI355X-native BLS12-381 G1 / MSM engine behind curdleproofs.pie's
compute_MSM / MSMAccumulator and the G1Point / Scalar surface of py_arkworks_bls12381.

Host code is plain Python + ctypes over libcurdle_g1.so (hand-written HIP for gfx950 + host C++);
no torch, no Triton, no CPU fallback for the batched paths.
"""
from .py_arkworks_bls12381 import G1Point, Scalar, CURVE_ORDER  # noqa: F401
from .msm_accumulator import MSMAccumulator, compute_MSM  # noqa: F401

__all__ = ["G1Point", "Scalar", "CURVE_ORDER", "MSMAccumulator", "compute_MSM"]
