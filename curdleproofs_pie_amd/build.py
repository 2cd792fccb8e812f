"""Build libcurdle_g1.so (HIP kernels for gfx950 + host C++ + C ABI) in-tree with the ROCm LLVM toolchain.

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
SOURCES = [os.path.join(CSRC, "msm_gpu.hip"), os.path.join(CSRC, "host_g1.cpp"), os.path.join(CSRC, "merlin.cpp"), os.path.join(CSRC, "shuffle_verify.cpp"), os.path.join(CSRC, "comm.cpp"), os.path.join(CSRC, "lazy_host.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, f) for f in ("kernels_opening.h", "fp28.h", "g1_xyzz.h", "g1_quad.h", "host_g1.h", "fe_mul_x86.h", "fr.h", "merlin_group.h", "bls_consts.h", "kernels_records.h", "kernels_prepare_digits.h",
                                                    "kernels_sort.h", "kernels_accumulate.h", "kernels_reduce.h", "kernels_small.h", "kernels_batch.h", "kernels_rows.h", "kernels_merlin.h", "kernels_frontend.h", "pool.h", "lazy_host.h", "fp_row.h", "glv.h", "host_context.h", "host_chains.h", "capi_core_msm.h", "capi_vec_batched.h", "capi_lincomb.h",
                                                    "capi_timing_batchmul.h", "capi_codec_transcripts.h", "capi_frontend.h", "capi_rows_probes.h")] + [
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
    h.update(os.environ.get("CURDLE_G1_PIPELINE", "staged").encode())
    with open(os.path.abspath(__file__), "rb") as f:      # the build recipe is part of what the library is built from
        h.update(f.read())
    return h.hexdigest()


HASH_FILE = LIB + ".srchash"
HOST_ONLY = ("host_g1.cpp", "merlin.cpp", "shuffle_verify.cpp", "comm.cpp", "lazy_host.cpp", "merlin_group.h", "fe_mul_x86.h")      # (pool.h and lazy_host.h are included by the .hip unit too)


def _hip_unit_hash(extra_flags=()) -> str:
    """Content hash of what the .hip translation unit is compiled from (every dependency that is not host-only)."""
    import hashlib

    h = hashlib.sha256()
    for d in sorted(DEPS):
        if os.path.basename(d) in HOST_ONLY or not os.path.exists(d):
            continue
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(repr(tuple(extra_flags)).encode())
    with open(os.path.abspath(__file__), "rb") as f:
        h.update(f.read())
    return h.hexdigest()[:24]


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(HASH_FILE):
        return True
    try:
        return open(HASH_FILE).read().strip() != _source_hash()
    except OSError:
        return True


LLVM_BIN = "/opt/rocm/lib/llvm/bin"
INFO_FILE = LIB + ".buildinfo"
# LLVM's Reassociate pass rewrites the carry chains of the 28-bit-limb Montgomery columns (fp28.h) into a form the
# AMDGPU backend then selects as one extra v_lshl_add_u64 per column: +26 VALU per field product, 4.7 % of k_accumulate
# (DESIGN.md section 9).  hipcc has no switch for a single pass, so the device side is built in explicit stages with
# the stock O3 pipeline minus that pass; every other pass, flag and device library is what hipcc itself uses
# (`hipcc -###`), and kernels without field arithmetic come out instruction-identical.
DROPPED_PASSES = ("reassociate",)


def default_pipeline() -> str:
    """"staged" (device code through opt without Reassociate) unless CURDLE_G1_PIPELINE=plain asks for one hipcc call."""
    p = os.environ.get("CURDLE_G1_PIPELINE", "staged")
    if p not in ("staged", "plain"):
        raise ValueError(f"CURDLE_G1_PIPELINE must be 'staged' or 'plain', not {p!r}")
    return p


def _run(cmd, verbose, **kw):
    if verbose:
        print("[curdleproofs_pie_amd.build]", " ".join(a if len(a) < 160 else a[:157] + "..." for a in cmd), flush=True)
    return subprocess.run(cmd, check=True, **kw)


def _device_pipeline(raw_bc: str, verbose: bool) -> str:
    """The textual O3 pipeline for gfx950 with DROPPED_PASSES removed and `internalize` told what hipcc's own
    (callback-built) internalize pass keeps: the kernels and every externally visible device variable."""
    import re

    tool = lambda t: os.path.join(LLVM_BIN, t)
    out = _run([tool("opt"), "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-amdgpu-internalize-symbols", "-passes=default<O3>", "-print-pipeline-passes",
                "-disable-output", raw_bc], verbose, capture_output=True, text=True).stdout.strip()
    pipe = out.replace(",BitcodeWriterPass", "")
    for dropped in DROPPED_PASSES:
        if f"{dropped}," not in pipe:
            raise RuntimeError(f"pass {dropped!r} not found in the O3 pipeline: the toolchain changed, re-check DESIGN.md section 9")
        pipe = pipe.replace(f"{dropped},", "")
    ll = _run([tool("llvm-dis"), raw_bc, "-o", "-"], False, capture_output=True, text=True).stdout
    keep = re.findall(r"^define [^@\n]*amdgpu_kernel [^@\n]*@([\w.$]+)\(", ll, re.M)
    keep += re.findall(r"^@([\w.$]+) = (?!internal|private)", ll, re.M)
    if not keep or pipe.count(",internalize,") != 1:
        raise RuntimeError("unexpected O3 pipeline text / no kernels found")
    return pipe.replace(",internalize,", ",internalize<%s>," % ";".join("preserve-gv=" + k for k in keep))


def _device_opt_bc(hip_src: str, td: str, common, verbose: bool) -> str:
    """raw device IR -> optimised device IR (td/dev_opt.bc)"""
    tool = lambda t: os.path.join(LLVM_BIN, t)
    j = lambda f: os.path.join(td, f)
    _run([tool("clang++"), "-x", "hip", "--offload-device-only", "--offload-arch=gfx950", *common, "-cuid=curdleg1", "-emit-llvm", "-Xclang", "-disable-llvm-passes",
          "-c", hip_src, "-o", j("dev_raw.bc")], verbose)
    pipe = _device_pipeline(j("dev_raw.bc"), verbose)
    _run([tool("opt"), "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", f"-passes={pipe}", j("dev_raw.bc"), "-o", j("dev_opt.bc")], verbose)
    return j("dev_opt.bc")


def emit_device_asm(out_path: str, extra_flags=(), verbose: bool = False) -> str:
    """The assembly listing (with the per-kernel register/scratch/occupancy comments tools/asm_stats.py reads) of exactly
    the device code the staged build ships."""
    import tempfile

    hip_src = [s for s in SOURCES if s.endswith(".hip")][0]
    with tempfile.TemporaryDirectory(prefix="curdle_g1_asm_") as td:
        bc = _device_opt_bc(hip_src, td, ["-O3", "-std=c++17", "-fPIC", *extra_flags], verbose)
        _run([os.path.join(LLVM_BIN, "llc"), "-O3", "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "--relocation-model=pic", "-filetype=asm", bc, "-o", out_path], verbose)
    return out_path


def _build_staged(out_path: str, extra_flags, verbose: bool) -> None:
    """clang (raw device IR) -> opt (O3 minus DROPPED_PASSES) -> llc -> lld -> offload bundle -> host compile embedding
    the bundle -> link.  The same steps `hipcc -###` shows, with `opt` made explicit."""
    import tempfile

    tool = lambda t: os.path.join(LLVM_BIN, t)
    for t in ("clang++", "opt", "llc", "lld", "llvm-dis", "clang-offload-bundler"):
        if not os.path.exists(tool(t)):
            raise RuntimeError(f"{tool(t)} not found")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    hip_src = [s for s in SOURCES if s.endswith(".hip")]
    cpp_src = [s for s in SOURCES if not s.endswith(".hip")]
    assert len(hip_src) == 1
    common = ["-O3", "-std=c++17", "-fPIC", *extra_flags]
    tgt = ["-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950"]
    # the .hip translation unit (device pipeline + its host half) is by far the slow part: keep its object keyed by the content
    # of everything it is built from, so a change to a host-only .cpp source relinks in seconds
    cache_dir = os.path.join(HERE, "..", "build", "hip_obj_cache")
    key = _hip_unit_hash(extra_flags)
    cached = os.path.join(cache_dir, f"hip_host_{key}.o")
    if os.path.exists(cached):
        _run([hipcc, *common, "-shared", cached, *cpp_src, "-ldl", "-o", out_path], verbose)
        return
    with tempfile.TemporaryDirectory(prefix="curdle_g1_build_") as td:
        j = lambda f: os.path.join(td, f)
        _device_opt_bc(hip_src[0], td, common, verbose)
        _run([tool("llc"), "-O3", *tgt, "--relocation-model=pic", "-filetype=obj", j("dev_opt.bc"), "-o", j("dev.o")], verbose)
        _run([tool("lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", j("dev.hsaco"), j("dev.o")], verbose)
        _run([tool("clang-offload-bundler"), "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
              "-input=/dev/null", f"-input={j('dev.hsaco')}", f"-output={j('dev.hipfb')}"], verbose)
        _run([tool("clang++"), "-x", "hip", "--offload-host-only", "--offload-arch=gfx950", *common, "-cuid=curdleg1", "-Xclang", "-fcuda-include-gpubinary", "-Xclang",
              j("dev.hipfb"), "-c", hip_src[0], "-o", j("hip_host.o")], verbose)
        _run([hipcc, *common, "-shared", j("hip_host.o"), *cpp_src, "-ldl", "-o", out_path], verbose)
        try:
            os.makedirs(cache_dir, exist_ok=True)
            for old in os.listdir(cache_dir):
                os.unlink(os.path.join(cache_dir, old))
            shutil.copyfile(j("hip_host.o"), cached + ".tmp")
            os.replace(cached + ".tmp", cached)
        except OSError:
            pass


def _build_plain(out_path: str, extra_flags, verbose: bool) -> None:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    _run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", *extra_flags, *SOURCES, "-ldl", "-o", out_path], verbose)


def _write_info(lib_path: str, pipeline: str, note: str = "") -> None:
    import json

    with open(lib_path + ".buildinfo", "w") as f:
        json.dump({"pipeline": pipeline, "dropped_passes": list(DROPPED_PASSES) if pipeline == "staged" else [], "note": note}, f)


def build_info(lib_path: str = LIB) -> dict:
    """How the library at lib_path was built ({"pipeline": "staged"|"plain", ...}); {} when unknown."""
    import json

    try:
        with open(lib_path + ".buildinfo") as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def build_variant(out_path: str, extra_flags, verbose: bool = True, pipeline: str | None = None) -> str:
    """A/B builds for experiments (extra -D flags, or pipeline="plain"); select at run time with CURDLE_G1_LIB=<path>."""
    pipeline = pipeline or default_pipeline()
    (_build_staged if pipeline == "staged" else _build_plain)(out_path, extra_flags, verbose)
    _write_info(out_path, pipeline)
    return out_path


PYFACE_SRC = os.path.join(CSRC, "pyface.c")


def pyface_path() -> str:
    import sysconfig

    return os.path.join(HERE, "_pyface" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_pyface(force: bool = False, verbose: bool = True) -> str:
    """The marshalling helper of the Python face (csrc/pyface.c: a CPython extension, plain gcc, no GPU code).  Optional: without
    it -- no compiler or no Python.h -- the Python face packs its lists in pure Python."""
    import hashlib
    import sysconfig

    out = pyface_path()
    with open(PYFACE_SRC, "rb") as f:
        want = hashlib.sha256(f.read()).hexdigest()
    stamp = out + ".srchash"
    if not force and os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return out
    inc = sysconfig.get_paths()["include"]
    if not os.path.exists(os.path.join(inc, "Python.h")):
        raise RuntimeError(f"Python.h not found under {inc}")
    tmp = f"{out}.tmp.{os.getpid()}"
    _run(["gcc", "-O2", "-fPIC", "-shared", "-pthread", "-Wall", f"-I{inc}", PYFACE_SRC, "-o", tmp], verbose)
    os.replace(tmp, out)
    with open(f"{stamp}.tmp.{os.getpid()}", "w") as f:      # ranks may build side by side: the stamp appears whole or not at all
        f.write(want)
    os.replace(f"{stamp}.tmp.{os.getpid()}", stamp)
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    """Build if the sources changed.  Safe under concurrent callers (e.g. 8 bench ranks on a fresh box): an flock
    serialises them, the staleness check is repeated under the lock and the .so is replaced atomically."""
    try:
        build_pyface(force, verbose)
    except (subprocess.CalledProcessError, RuntimeError, OSError) as e:
        print(f"[curdleproofs_pie_amd.build] note: _pyface not built ({e}); the Python face packs its lists in pure Python", file=sys.stderr, flush=True)
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
            pipeline, note = default_pipeline(), ""
            if pipeline == "staged":
                try:
                    _build_staged(tmp, [], verbose)
                except (subprocess.CalledProcessError, RuntimeError, OSError) as e:
                    # same sources, same results, ~5 % more VALU work in the EC kernels: say so and carry on
                    note = f"staged pipeline failed ({e}); built with one hipcc call instead"
                    print(f"[curdleproofs_pie_amd.build] WARNING: {note}", file=sys.stderr, flush=True)
                    pipeline = "plain"
            if pipeline == "plain":
                _build_plain(tmp, [], verbose)
            os.replace(tmp, LIB)
            _write_info(LIB, pipeline, note)
            with open(HASH_FILE + f".{os.getpid()}", "w") as f:
                f.write(_source_hash())
            os.replace(HASH_FILE + f".{os.getpid()}", HASH_FILE)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
