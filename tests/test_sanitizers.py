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
