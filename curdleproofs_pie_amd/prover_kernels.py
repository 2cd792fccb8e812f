"""Prover-side hot loops of the shuffle argument, re-cut for the GPU.

The reference prover spends its time in four patterns, all written as Python loops over single `G1Point` operators:

  * the halving rounds of the inner-product argument       ipa.py:117-151      4 MSMs of h terms + 2 folds of h points per round
  * the halving rounds of the same-MSM argument            same_msm.py:93-130  6 MSMs of h terms + 3 folds of h points per round
  * `shuffle_permute_and_commit_input`                     curdleproofs.py:301-321   2 ell same-scalar multiplications + 2 MSMs
  * the grand-product base change G'_i = G_i * beta^-(i+1) grand_prod.py:64-71 ell + 4 per-index multiplications

Here every round is ONE batched GPU MSM call (regime B: `compute_MSM_batch`) plus ONE batched fold (`batch_fold_scalars`:
out[i] = L[i] + gamma_i * R[i], k_batch_mul), and the map patterns are one `batch_mul*` launch each.  The `*_many` forms run
several independent provers in step: round k of ALL of them is still one MSM call and one fold launch (cross-proof batching;
tools/gpu_prover_bench.py measures both).  The Fiat-Shamir
transcript stays with the caller, exactly where the reference has it: the round functions take a `next_gamma` callback that
receives the round's commitments (to absorb them) and returns the challenge.  Scalars follow the reference's update order, so
the outputs are the same group elements / field elements the reference prover produces (tests/test_prover_kernels_gpu.py
replays inputs recorded from the reference prover and compares every L / R point and final scalar).
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

from .msm_accumulator import batch_fold_scalars, batch_mul, batch_mul_same_scalar, compute_MSM, compute_MSM_batch
from .py_arkworks_bls12381 import G1Point, Scalar
from .util import random_scalar

N_BLINDERS = 4                                                          # curdleproofs.py:24


def _inner(a: Sequence[Scalar], b: Sequence[Scalar]) -> Scalar:          # util.py:85-87
    acc = 0
    for x, y in zip(a, b):
        acc += x._v * y._v
    return Scalar(acc)


def ipa_rounds_many(provers: Sequence[Tuple[Sequence[G1Point], Sequence[G1Point], G1Point, Sequence[Scalar], Sequence[Scalar]]],
                    next_gammas: Sequence[Callable[[G1Point, G1Point, G1Point, G1Point], Scalar]]):
    """ipa.py:117-151 for SEVERAL independent provers in step (same vector length): round k of all of them is ONE regime-B MSM
    call (4 MSMs per prover) and ONE fold launch -- the cross-proof batched form.  provers[p] = (crs_G_vec, crs_G_prime_vec, H,
    vec_c, vec_d) with `H` = crs_H * beta (ipa.py:110) and the vectors already blinded (ipa.py:107-109).
    -> per prover (vec_L_C, vec_R_C, vec_L_D, vec_R_D, c_final, d_final)."""
    st = [dict(G=list(G), Gp=list(Gp), H=H, c=list(c), d=list(d), LC=[], RC=[], LD=[], RD=[]) for G, Gp, H, c, d in provers]
    n = len(st[0]["c"])
    assert all(len(s["c"]) == len(s["d"]) == len(s["G"]) == len(s["Gp"]) == n for s in st) and n & (n - 1) == 0
    while n > 1:
        n //= 2
        jobs = []
        for s in st:
            c_L, c_R, d_L, d_R = s["c"][:n], s["c"][n:], s["d"][:n], s["d"][n:]
            # L_C = MSM(G_R, c_L) + H <c_L, d_R>;  L_D = MSM(G'_L, d_R);  R_C = MSM(G_L, c_R) + H <c_R, d_L>;  R_D = MSM(G'_R, d_L)
            jobs += [(s["G"][n:] + [s["H"]], c_L + [_inner(c_L, d_R)]), (s["Gp"][:n], d_R),
                     (s["G"][:n] + [s["H"]], c_R + [_inner(c_R, d_L)]), (s["Gp"][n:], d_L)]
        res = compute_MSM_batch(jobs)
        lefts, rights, scal = [], [], []
        for i, (s, ng) in enumerate(zip(st, next_gammas)):
            L_C, L_D, R_C, R_D = res[4 * i: 4 * i + 4]
            s["LC"].append(L_C); s["RC"].append(R_C); s["LD"].append(L_D); s["RD"].append(R_D)
            gamma = ng(L_C, L_D, R_C, R_D)
            gamma_inv = gamma.inverse()
            s["c"] = [l + gamma_inv * r for l, r in zip(s["c"][:n], s["c"][n:])]
            s["d"] = [l + gamma * r for l, r in zip(s["d"][:n], s["d"][n:])]
            lefts += s["G"][:n] + s["Gp"][:n]                            # G_L[i] + G_R[i] * gamma,  G'_L[i] + G'_R[i] * gamma^-1
            rights += s["G"][n:] + s["Gp"][n:]
            scal += [gamma] * n + [gamma_inv] * n
        folded = batch_fold_scalars(lefts, rights, scal)
        for i, s in enumerate(st):
            s["G"], s["Gp"] = folded[2 * n * i: 2 * n * i + n], folded[2 * n * i + n: 2 * n * (i + 1)]
    return [(s["LC"], s["RC"], s["LD"], s["RD"], s["c"][0], s["d"][0]) for s in st]


def ipa_rounds(crs_G_vec: Sequence[G1Point], crs_G_prime_vec: Sequence[G1Point], H: G1Point, vec_c: Sequence[Scalar],
               vec_d: Sequence[Scalar], next_gamma: Callable[[G1Point, G1Point, G1Point, G1Point], Scalar]):
    """ipa.py:117-151.  `H` is crs_H * beta (ipa.py:110); vec_c / vec_d are the blinded vectors (after ipa.py:107-109).
    -> (vec_L_C, vec_R_C, vec_L_D, vec_R_D, c_final, d_final)."""
    return ipa_rounds_many([(crs_G_vec, crs_G_prime_vec, H, vec_c, vec_d)], [next_gamma])[0]


def same_msm_rounds_many(provers: Sequence[Tuple[Sequence[G1Point], Sequence[G1Point], Sequence[G1Point], Sequence[Scalar]]],
                         next_gammas: Sequence[Callable[..., Scalar]]):
    """same_msm.py:93-130 for several independent provers in step: per round ONE regime-B MSM call (6 MSMs per prover) and ONE
    fold launch.  provers[p] = (crs_G_vec, vec_T, vec_U, vec_x) with vec_x already blinded (:89-91).
    -> per prover (vec_L_A, vec_L_T, vec_L_U, vec_R_A, vec_R_T, vec_R_U, x_final)."""
    st = [dict(G=list(G), T=list(T), U=list(U), x=list(x), out=[[] for _ in range(6)]) for G, T, U, x in provers]
    n = len(st[0]["x"])
    assert all(len(s["x"]) == len(s["G"]) == len(s["T"]) == len(s["U"]) == n for s in st) and n & (n - 1) == 0
    while n > 1:
        n //= 2
        jobs = []
        for s in st:
            x_L, x_R = s["x"][:n], s["x"][n:]
            jobs += [(s["G"][n:], x_L), (s["T"][n:], x_L), (s["U"][n:], x_L), (s["G"][:n], x_R), (s["T"][:n], x_R), (s["U"][:n], x_R)]    # L_A L_T L_U R_A R_T R_U
        res = compute_MSM_batch(jobs)
        lefts, rights, scal = [], [], []
        for i, (s, ng) in enumerate(zip(st, next_gammas)):
            rnd = res[6 * i: 6 * i + 6]
            for lst, pnt in zip(s["out"], rnd):
                lst.append(pnt)
            gamma = ng(*rnd)
            gamma_inv = gamma.inverse()
            s["x"] = [l + gamma_inv * r for l, r in zip(s["x"][:n], s["x"][n:])]
            lefts += s["T"][:n] + s["U"][:n] + s["G"][:n]                # the three folds of a prover share gamma
            rights += s["T"][n:] + s["U"][n:] + s["G"][n:]
            scal += [gamma] * (3 * n)
        folded = batch_fold_scalars(lefts, rights, scal)
        for i, s in enumerate(st):
            f = folded[3 * n * i: 3 * n * (i + 1)]
            s["T"], s["U"], s["G"] = f[:n], f[n: 2 * n], f[2 * n:]
    return [(*s["out"], s["x"][0]) for s in st]


def same_msm_rounds(crs_G_vec: Sequence[G1Point], vec_T: Sequence[G1Point], vec_U: Sequence[G1Point], vec_x: Sequence[Scalar],
                    next_gamma: Callable[[G1Point, G1Point, G1Point, G1Point, G1Point, G1Point], Scalar]):
    """same_msm.py:93-130 (vec_x already blinded, :89-91).  -> (vec_L_A, vec_L_T, vec_L_U, vec_R_A, vec_R_T, vec_R_U, x_final)."""
    return same_msm_rounds_many([(crs_G_vec, vec_T, vec_U, vec_x)], [next_gamma])[0]


def shuffle_permute_and_commit_input(crs, vec_R: Sequence[G1Point], vec_S: Sequence[G1Point], permutation: Sequence[int], k: Scalar
                                     ) -> Tuple[List[G1Point], List[G1Point], G1Point, List[Scalar]]:
    """Drop-in for curdleproofs.py:301-321: vec_T = perm([R * k]), vec_U = perm([S * k]) as one same-scalar launch, M as one GPU
    MSM over vec_G | vec_H.  Draws the N_BLINDERS blinders exactly where the reference does (util.py:81-82)."""
    ell = len(crs.vec_G)
    both = batch_mul_same_scalar(list(vec_R) + list(vec_S), k)
    vec_T = [both[j] for j in permutation]                               # get_permutation, util.py:93-96
    vec_U = [both[len(vec_R) + j] for j in permutation]
    sigma_ell = [Scalar(j) for j in permutation]
    vec_m_blinders = [random_scalar() for _ in range(N_BLINDERS)]
    M = compute_MSM(list(crs.vec_G) + list(crs.vec_H), sigma_ell + vec_m_blinders)
    assert len(sigma_ell) == ell
    return vec_T, vec_U, M, vec_m_blinders


def grand_product_bases(crs_G_vec: Sequence[G1Point], crs_H_vec: Sequence[G1Point], beta_inv: Scalar) -> Tuple[List[G1Point], List[G1Point]]:
    """grand_prod.py:64-71: G'_i = G_i * beta^-(i+1), H'_i = H_i * beta^-(ell+1), one per-index launch."""
    ell = len(crs_G_vec)
    pows, p = [], beta_inv
    for _ in range(ell):
        pows.append(p)
        p = p * beta_inv
    res = batch_mul(list(crs_G_vec) + list(crs_H_vec), pows + [p] * len(crs_H_vec))
    return res[:ell], res[ell:]
