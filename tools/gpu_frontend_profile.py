#!/usr/bin/env python3
"""Where the native front-end's time goes (needs a library built with -DCG1_FE_PROFILE; host only, no GPU work)."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from curdleproofs_pie_amd import build as B
lib = os.path.join(ROOT, "build", "libcurdle_g1_prof.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
B.build_variant(lib, ["-DCG1_FE_PROFILE"], verbose=False)
os.environ["CURDLE_G1_LIB"] = lib
from curdleproofs_pie_amd import _native as N
N.tune_runtime()
from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier
case = [c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "shuffle_vectors.json")))["cases"] if c["ell"] == 124][0]
v = ShuffleBatchVerifier(bytes.fromhex(case["crs"]), threads=1)
n = 256
inst = bytes.fromhex(case["pre_r"] + case["pre_k"] + case["post_r"] + case["post_k"]) * n
proofs = bytes.fromhex(case["proof"]) * n
w = v.draw_weights(n)
# decoded window from the host decoder (so the front-end skips its own square roots, as in the GPU flow)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle.shuffle_check import decompress_affine
L = v.crs.points_per_proof
wire = ctypes.create_string_buffer(L * 48)
N.cg1_shuffle_gather_points(v.crs.handle, 1, inst[: 4 * 124 * 48], proofs[: v.crs.proof_bytes], wire)
dec, _ = decompress_affine(wire.raw[(4 * 124 + 1) * 48: (4 * 124 + 9) * 48], 8)
prof = N.lib.cg1_shuffle_profile
out = (ctypes.c_double * 6)()
v.prepare(inst, proofs, n, weights=w, decoded=dec * n)
prof(out)
N.lib.cg1_transcript_profile((ctypes.c_double * 4)())
t0 = time.perf_counter(); v.prepare(inst, proofs, n, weights=w, decoded=dec * n); dt = time.perf_counter() - t0
prof(out)
tp = (ctypes.c_double * 4)()
N.lib.cg1_transcript_profile(tp)
print(f"  transcript detail: {tp[3] / n:.0f} x8 permutation calls per proof; per proof: gather {1e3 * tp[0] / n:.1f} us, Keccak x8 {1e3 * tp[1] / n:.1f} us, scatter {1e3 * tp[2] / n:.1f} us (incl. ~25 ns of timer overhead per call and part)")
names = ["parse+decode", "transcript", "fr between", "ec (D, A')", "scalar rows", "-"]
print(f"front-end, one thread: {1e6 * dt / n:.1f} us per proof")
for nm, x in zip(names, out):
    if nm != "-":
        print(f"  {nm:14s} {1e3 * x / n:7.1f} us per proof")
