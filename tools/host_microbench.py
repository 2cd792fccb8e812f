#!/usr/bin/env python3
"""Host-side primitive costs behind the shuffle front-end (run on the box whose numbers you want)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N

st = ctypes.create_string_buffer(208)
N.cg1_merlin_init(st, b"curdleproofs", 12)
items = bytes(range(48)) * 1000
t = time.perf_counter()
for _ in range(50):
    N.cg1_merlin_append_list(st, b"curdleproofs_step1", 18, items, 48, 1000)
dt = (time.perf_counter() - t) / 50
print(f"merlin append 48 B message: {dt * 1e9 / 1000:.0f} ns  (70 B absorbed -> {dt * 1e9 / 1000 / (70 / 166):.0f} ns per Keccak-f)")
out = ctypes.create_string_buffer(32)
t = time.perf_counter()
for _ in range(5000):
    N.cg1_merlin_challenge_scalar(st, b"curdleproofs_vec_a", 18, out)
print(f"challenge_scalar: {(time.perf_counter() - t) / 5000 * 1e6:.2f} us")
g = ctypes.create_string_buffer(144); N.cg1_generator(g)
c = ctypes.create_string_buffer(48); N.cg1_compress(c, g.raw)
blob = ctypes.create_string_buffer(144)
t = time.perf_counter()
for _ in range(5000):
    N.cg1_decompress(blob, c.raw, 0)
print(f"host decompress (sqrt): {(time.perf_counter() - t) / 5000 * 1e6:.1f} us")
t = time.perf_counter()
for _ in range(5000):
    N.cg1_compress(c, blob.raw)
print(f"host compress (inversion): {(time.perf_counter() - t) / 5000 * 1e6:.1f} us")
k = (123456789 ** 7 % (1 << 250)).to_bytes(32, "little")
t = time.perf_counter()
for _ in range(2000):
    N.cg1_mul(blob, g.raw, k)
print(f"host scalar mul (4-bit windows): {(time.perf_counter() - t) / 2000 * 1e6:.1f} us")
lanes = (ctypes.c_uint64 * 200)()
t = time.perf_counter()
for _ in range(20000):
    N.cg1_keccak_f1600_x8(lanes)
dt8 = (time.perf_counter() - t) / 20000
st200 = ctypes.create_string_buffer(200)
t = time.perf_counter()
for _ in range(20000):
    N.cg1_keccak_f1600(st200)
dt1 = (time.perf_counter() - t) / 20000
print(f"Keccak-f[1600]: one state {dt1 * 1e9:.0f} ns per call; eight states (AVX-512 when present) {dt8 * 1e9:.0f} ns per call = {dt8 * 1e9 / 8:.0f} ns per state (both incl. ~0.2 us of ctypes call overhead)")
print("usable host threads (affinity mask capped by cgroup quota):", N.cg1_shuffle_default_threads(), " os.cpu_count:", os.cpu_count())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.__class__.__name__)
