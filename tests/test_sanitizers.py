"""Host-side C++ of the product under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; GPU
sanitizers are not available on the pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_under_asan_ubsan():
    out_dir = os.path.join(ROOT, "tests", "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "sanitize_driver")
    src = [os.path.join(ROOT, "tests", "native", "sanitize_driver.cpp"),
           os.path.join(ROOT, "curdleproofs_pie_amd", "csrc", "host_g1.cpp"),
           os.path.join(ROOT, "curdleproofs_pie_amd", "csrc", "merlin.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", *src, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                       env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "sanitize ok" in r.stdout


def _shuffle_case_blob(tmp_path):
    import json
    import struct

    with open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")) as f:
        case = json.load(f)["cases"][1]
    blob = tmp_path / "case.bin"
    blob.write_bytes(struct.pack("<Q", case["ell"]) + bytes.fromhex(case["crs"]) +
                     bytes.fromhex(case["pre_r"] + case["pre_k"] + case["post_r"] + case["post_k"]) + bytes.fromhex(case["proof"]))
    return blob


def test_shuffle_front_end_under_tsan(tmp_path):
    """Race detection: the native worker pool + grouped transcripts under ThreadSanitizer (same driver, 1 and 4 threads)."""
    blob = _shuffle_case_blob(tmp_path)
    out_dir = os.path.join(ROOT, "tests", "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "tsan_shuffle")
    csrc = os.path.join(ROOT, "curdleproofs_pie_amd", "csrc")
    src = [os.path.join(ROOT, "tests", "native", "sanitize_shuffle.cpp"), os.path.join(csrc, "shuffle_verify.cpp"),
           os.path.join(csrc, "host_g1.cpp"), os.path.join(csrc, "merlin.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-fno-omit-frame-pointer", "-Wno-psabi",
                           *src, "-o", exe])
    r = subprocess.run([exe, str(blob)], capture_output=True, text=True, timeout=600,
                       env={**os.environ, "TSAN_OPTIONS": "halt_on_error=1"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "sanitize ok" in r.stdout


def test_shuffle_front_end_under_asan_ubsan(tmp_path):
    """csrc/shuffle_verify.cpp (+ fr.h, host_g1.cpp, merlin.cpp): golden proof, bit flips and garbage, 1 and 4 threads."""
    import json
    import struct

    with open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")) as f:
        case = json.load(f)["cases"][1]
    blob = tmp_path / "case.bin"
    blob.write_bytes(struct.pack("<Q", case["ell"]) + bytes.fromhex(case["crs"]) +
                     bytes.fromhex(case["pre_r"] + case["pre_k"] + case["post_r"] + case["post_k"]) + bytes.fromhex(case["proof"]))
    out_dir = os.path.join(ROOT, "tests", "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "sanitize_shuffle")
    csrc = os.path.join(ROOT, "curdleproofs_pie_amd", "csrc")
    src = [os.path.join(ROOT, "tests", "native", "sanitize_shuffle.cpp"), os.path.join(csrc, "shuffle_verify.cpp"),
           os.path.join(csrc, "host_g1.cpp"), os.path.join(csrc, "merlin.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", *src, "-o", exe])
    r = subprocess.run([exe, str(blob)], capture_output=True, text=True, timeout=600,
                       env={**os.environ, "ASAN_OPTIONS": "detect_leaks=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "sanitize ok" in r.stdout


import pytest


@pytest.mark.parametrize("flags", [["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], ["-fsanitize=thread"]], ids=["asan_ubsan", "tsan"])
def test_comm_socket_transport_under_sanitizers(flags):
    """csrc/comm.cpp: four ranks as threads over the TCP control channel (rendezvous with a stranger on the port, all-gathers,
    the G1 all-reduce, a mismatched collective) under ASan + UBSan and under ThreadSanitizer."""
    out_dir = os.path.join(ROOT, "tests", "native", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "sanitize_comm_" + ("tsan" if "thread" in flags[0] else "asan"))
    csrc = os.path.join(ROOT, "curdleproofs_pie_amd", "csrc")
    src = [os.path.join(ROOT, "tests", "native", "sanitize_comm.cpp"), os.path.join(csrc, "comm.cpp"), os.path.join(csrc, "host_g1.cpp")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-pthread", *flags, "-fno-omit-frame-pointer", "-Wno-psabi", "-I/opt/rocm/include",
                           *src, "-ldl", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                       env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "print_stacktrace=1", "TSAN_OPTIONS": "halt_on_error=1"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "sanitize ok" in r.stdout


def test_threaded_list_walks_under_tsan(tmp_path):
    """csrc/pyface.c built with ThreadSanitizer and loaded into this interpreter's twin (libtsan preloaded): the helper crew of the long
    walks (pack_points / pack_scalars over 2^17 objects, whole and in slices, 6 threads) races with nothing -- the objects are immutable,
    the caller holds the GIL, every thread writes its own part of the destination."""
    import sys
    import sysconfig

    inc = sysconfig.get_paths()["include"]
    if not os.path.exists(os.path.join(inc, "Python.h")):
        pytest.skip("Python.h not available")
    tsan = subprocess.run(["gcc", "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    if not tsan or not os.path.isabs(tsan) or not os.path.exists(tsan):
        pytest.skip("libtsan not available")
    so = tmp_path / "_pyface.so"
    subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-shared", "-pthread", "-fsanitize=thread", f"-I{inc}",
                           os.path.join(ROOT, "curdleproofs_pie_amd", "csrc", "pyface.c"), "-o", str(so)])
    script = tmp_path / "walk.py"
    script.write_text('''
import sys, ctypes
sys.path.insert(0, %r)
import _pyface as P
class Sc:
    __slots__ = ("_v",)
class Pt:
    __slots__ = ("_blob", "_a", "_k", "_t", "_sg", "_seq", "__weakref__")
P.bind(Pt, Sc, [], bytes(144))
n = 1 << 17
pts = []
for i in range(n):
    p = Pt(); p._blob = bytes([i & 255]) * 144; p._a = None; p._k = None; p._t = None; p._sg = None; p._seq = None
    pts.append(p)
scs = []
for i in range(n):
    s = Sc(); s._v = (i * 0x9E3779B97F4A7C15) & ((1 << 255) - 1); scs.append(s)
buf = ctypes.create_string_buffer(144 * n); sb = ctypes.create_string_buffer(32 * n)
P.set_threads(6)
for rep in range(4):
    assert P.pack_points(pts, ctypes.addressof(buf), n)[0] == n
    assert P.pack_scalars(scs, ctypes.addressof(sb), n) == n
    for off in range(0, n, 1 << 15):
        P.pack_points(pts, ctypes.addressof(buf) + 144 * off, 1 << 15, off, 1 << 15)
        P.pack_scalars(scs, ctypes.addressof(sb) + 32 * off, 1 << 15, off, 1 << 15)
assert buf.raw == b"".join(p._blob for p in pts)
assert sb.raw == b"".join(s._v.to_bytes(32, "little") for s in scs)
scs[77777] = -5                       # an element the digit reader refuses, found by a helper thread, raised by the caller
try:
    P.pack_scalars(scs, ctypes.addressof(sb), n)
    raise SystemExit("no OverflowError")
except OverflowError:
    pass
pts[100001]._blob = None              # a deferred value in the middle of the list
try:
    P.pack_points(pts, ctypes.addressof(buf), n)
    raise SystemExit("no Unforced")
except P.Unforced:
    pass
print("walk ok")
''' % str(tmp_path))
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900,
                       env={**os.environ, "LD_PRELOAD": tsan, "TSAN_OPTIONS": "halt_on_error=1 report_signal_unsafe=0"})
    assert r.returncode == 0 and "walk ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
