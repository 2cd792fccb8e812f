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
