#!/usr/bin/env python3
"""Prover-side flows at ell = 124 (vectors of 128) on the GPU: how long the reference prover's hot loops take when each halving
round is one regime-B MSM call + one fold launch (curdleproofs_pie_amd/prover_kernels.py), for one prover and for B provers in
step (cross-proof batching: round k of all B provers in ONE MSM call and ONE fold launch).

Reference loops: ipa.py:117-151 (7 rounds: 4 MSMs + 2 folds each), same_msm.py:93-130 (7 rounds: 6 MSMs + 3 folds),
curdleproofs.py:301-321 (2 ell same-scalar multiplications + one MSM), grand_prod.py:64-71 (ell + 4 per-index multiplications).
Inputs are synthetic (random points k_i * G, random scalars); the Fiat-Shamir callback returns fixed challenges -- the
transcript is the caller's (host) business and is not timed here.  Reports wall time per flow (Python marshalling included: the
flows take and return G1Point objects, as the reference's do) and, beside it, the GPU time of the MSM / fold calls alone."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from curdleproofs_pie_amd import _native as N
from curdleproofs_pie_amd import msm_accumulator as M
from curdleproofs_pie_amd import prover_kernels as K
from curdleproofs_pie_amd.py_arkworks_bls12381 import G1Point, Scalar

ELL, NB = 124, 4
n = ELL + NB
rng = random.Random(7)
ctx = N.default_context()
R = int("73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001", 16)
rs = lambda: Scalar(rng.randint(1, R - 1))
pts = lambda m: M.batch_mul([G1Point()] * m, [rs() for _ in range(m)])

# GPU-side time of the two device entry points the flows use, accumulated through wrappers
gpu = {"msm_ms": 0.0, "msm_calls": 0, "fold_ms": 0.0, "fold_calls": 0}
_msm, _fold = ctx.msm_batched_host, ctx.batch_mul_add_host


def msm_timed(*a):
    t = time.perf_counter(); out = _msm(*a); gpu["msm_ms"] += (time.perf_counter() - t) * 1e3; gpu["msm_calls"] += 1
    return out


def fold_timed(*a):
    t = time.perf_counter(); out = _fold(*a); gpu["fold_ms"] += (time.perf_counter() - t) * 1e3; gpu["fold_calls"] += 1
    return out


ctx.msm_batched_host, ctx.batch_mul_add_host = msm_timed, fold_timed
gam = [rs() for _ in range(16)]
const = lambda: (lambda g: (lambda *p: g.pop(0)))(list(gam))


def timed(label, fn, reps=3):
    fn()
    best = None
    for _ in range(reps):
        for k in gpu:
            gpu[k] = 0
        t = time.perf_counter(); fn(); dt = (time.perf_counter() - t) * 1e3
        if best is None or dt < best[0]:
            best = (dt, dict(gpu))
    dt, g = best
    print("%-58s %8.2f ms wall | device calls: %2d MSM %7.2f ms, %2d fold/map %7.2f ms" % (label, dt, g["msm_calls"], g["msm_ms"], g["fold_calls"], g["fold_ms"]), flush=True)
    return dt


G, Gp, T, U = pts(n), pts(n), pts(n), pts(n)
H = pts(1)[0]
c, d, x = [rs() for _ in range(n)], [rs() for _ in range(n)], [rs() for _ in range(n)]
print("vectors of %d (ell = %d + %d blinders); times are best of 3" % (n, ELL, NB))
t_ipa = timed("IPA halving rounds, 1 prover (ipa.py:117-151)", lambda: K.ipa_rounds(G, Gp, H, c, d, const()))
t_sm = timed("same-MSM halving rounds, 1 prover (same_msm.py:93-130)", lambda: K.same_msm_rounds(G, T, U, x, const()))


class Crs:
    vec_G, vec_H = G[:ELL], G[ELL:]


perm = list(range(ELL)); rng.shuffle(perm)
t_pc = timed("shuffle_permute_and_commit_input (curdleproofs.py:301-321)", lambda: K.shuffle_permute_and_commit_input(Crs, T[:ELL], U[:ELL], perm, rs()))
t_gp = timed("grand-product base change (grand_prod.py:64-71)", lambda: K.grand_product_bases(G[:ELL], G[ELL:], rs()))
print("GPU-side flows of ONE ell = 124 proof: %.1f ms wall (the reference's own loops over the host backend: seconds)" % (t_ipa + t_sm + t_pc + t_gp))
coeffs = K.grand_product_coeffs(ELL, NB, rs())
t_ipa2 = timed("IPA rounds with the base change passed as coefficients (no G' made)", lambda: K.ipa_rounds(G, G, H, c, d, const(), G_prime_coeffs=coeffs))
print("... with the implicit base change: %.1f ms" % (t_ipa2 + t_sm + t_pc))
for B in (8, 64):
    ip = [(G, Gp, H, [rs() for _ in range(n)], [rs() for _ in range(n)]) for _ in range(B)]
    sm = [(G, T, U, [rs() for _ in range(n)]) for _ in range(B)]
    a = timed("IPA rounds, %d provers in step (cross-proof batched)" % B, lambda: K.ipa_rounds_many(ip, [const() for _ in range(B)]), reps=2)
    b = timed("same-MSM rounds, %d provers in step" % B, lambda: K.same_msm_rounds_many(sm, [const() for _ in range(B)]), reps=2)
    print("   -> per prover: IPA %.2f ms, same-MSM %.2f ms" % (a / B, b / B))
