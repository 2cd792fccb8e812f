"""Build libcurdle_g1.so (HIP kernels for gfx950 + host C++ + C ABI) in-tree with hipcc.

    python -m curdleproofs_pie_amd.build          # or: from curdleproofs_pie_amd.build import build; build()

hipcc cross-compiles gfx950 without a GPU.  The .so is git-ignored but travels to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcurdle_g1.so")
SOURCES = [os.path.join(CSRC, "msm_gpu.hip"), os.path.join(CSRC, "host_g1.cpp"), os.path.join(CSRC, "merlin.cpp"), os.path.join(CSRC, "shuffle_verify.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in ("fp28.h", "g1_xyzz.h", "g1_quad.h", "host_g1.h", "fr.h", "merlin_group.h", "bls_consts.h", "kernels_records.h", "kernels_prepare_digits.h",
                                                    "kernels_sort.h", "kernels_accumulate.h", "kernels_reduce.h", "kernels_batch.h", "kernels_rows.h", "kernels_merlin.h")] + [
    os.path.join(HERE, "..", "include", "curdle_g1.h")
]


def _source_hash() -> str:
    """Content hash of every source the library is built from (mtimes do not survive copying the tree)."""
    import hashlib

    h = hashlib.sha256()
    for d in sorted(DEPS):
        if os.path.exists(d):
            h.update(os.path.basename(d).encode())
            with open(d, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


HASH_FILE = LIB + ".srchash"


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(HASH_FILE):
        return True
    try:
        return open(HASH_FILE).read().strip() != _source_hash()
    except OSError:
        return True


def build_variant(out_path: str, extra_flags, verbose: bool = True) -> str:
    """A/B builds for experiments (e.g. -DCG1_NO_ASM_MAD); select at run time with CURDLE_G1_LIB=<path>."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", *extra_flags, *SOURCES, "-o", out_path]
    if verbose:
        print("[curdleproofs_pie_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out_path


def build(force: bool = False, verbose: bool = True) -> str:
    """Build if the sources changed.  Safe under concurrent callers (e.g. 8 bench ranks on a fresh box): an flock
    serialises them, the staleness check is repeated under the lock and the .so is replaced atomically."""
    if not force and not needs_build():
        return LIB
    import fcntl

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libcurdle_g1.so (HIP toolchain required)")
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            tmp = f"{LIB}.tmp.{os.getpid()}"
            cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", *SOURCES, "-o", tmp]
            if verbose:
                print("[curdleproofs_pie_amd.build]", " ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
            with open(HASH_FILE + f".{os.getpid()}", "w") as f:
                f.write(_source_hash())
            os.replace(HASH_FILE + f".{os.getpid()}", HASH_FILE)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
