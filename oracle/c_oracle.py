"""ctypes loader for the plain-C CPU oracle (oracle/msm_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product."""
from __future__ import annotations

import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libmsm_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "msm_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B" if force else "-s", "all"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        _lib.orc_compute_msm.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
        _lib.orc_msm_bucket.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        _lib.orc_scalar_mul.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p]
        _lib.orc_add.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_void_p]
        _lib.orc_compress.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
        _lib.orc_msm_bucket_mt.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _lib.orc_decompress.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p]
        _lib.orc_decompress.restype = ctypes.c_int
    return _lib


def compute_msm(points96: bytes, scalars32: bytes, n: int) -> bytes:
    """msm_accumulator.py:6-12 on affine96 / scalar32 buffers; returns affine96 (zeros = identity)."""
    out = ctypes.create_string_buffer(96)
    lib().orc_compute_msm(points96, scalars32, n, out)
    return out.raw


def msm_bucket(points96: bytes, scalars32: bytes, n: int, c: int = 0) -> bytes:
    """Bucket-method MSM (same result as compute_msm, NOT the reference's algorithm; for large parity sizes)."""
    if c <= 0:
        c = max(4, min(16, n.bit_length() - 3))
    out = ctypes.create_string_buffer(96)
    lib().orc_msm_bucket(points96, scalars32, n, c, out)
    return out.raw


def scalar_mul(point96: bytes, scalar32: bytes) -> bytes:
    out = ctypes.create_string_buffer(96)
    lib().orc_scalar_mul(point96, scalar32, out)
    return out.raw


def add(a96: bytes, b96: bytes) -> bytes:
    out = ctypes.create_string_buffer(96)
    lib().orc_add(a96, b96, out)
    return out.raw


def compress(a96: bytes) -> bytes:
    out = ctypes.create_string_buffer(48)
    lib().orc_compress(a96, out)
    return out.raw


def msm_bucket_mt(points96: bytes, scalars32: bytes, n: int, threads: int, c: int = 0) -> bytes:
    """msm_bucket on `threads` host cores (OpenMP).  A stronger NON-reference baseline (the reference is single-threaded)."""
    if c <= 0:
        c = max(4, min(16, n.bit_length() - 3))
    out = ctypes.create_string_buffer(96)
    lib().orc_msm_bucket_mt(points96, scalars32, n, c, threads, out)
    return out.raw


def decompress(data48: bytes, check_subgroup: bool = False):
    """-> (status, affine96): G1Point.from_compressed_bytes[_unchecked] (util.py:35-36); status 0 ok, 1 bad encoding,
    2 not on the curve, 3 not in the subgroup."""
    out = ctypes.create_string_buffer(96)
    rc = lib().orc_decompress(bytes(data48), 1 if check_subgroup else 0, out)
    return rc, out.raw
