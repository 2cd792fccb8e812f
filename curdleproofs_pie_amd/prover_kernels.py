"""Prover-side hot loops of the shuffle argument, re-cut for the GPU.

The reference prover spends its time in four patterns, all written as Python loops over single `G1Point` operators:

  * the halving rounds of the inner-product argument       ipa.py:117-151      4 MSMs of h terms + 2 folds of h points per round
  * the halving rounds of the same-MSM argument            same_msm.py:93-130  6 MSMs of h terms + 3 folds of h points per round
  * `shuffle_permute_and_commit_input`                     curdleproofs.py:301-321   2 ell same-scalar multiplications + 2 MSMs
  * the grand-product base change G'_i = G_i * beta^-(i+1) grand_prod.py:64-71 ell + 4 per-index multiplications

Here every round is ONE batched GPU MSM call (regime B: `compute_MSM_batch`) plus ONE batched fold (`batch_fold`:
out[i] = L[i] + gamma * R[i], k_batch_mul), and the map patterns are one `batch_mul*` launch each.  The Fiat-Shamir
transcript stays with the caller, exactly where the reference has it: the round functions take a `next_gamma` callback that
receives the round's commitments (to absorb them) and returns the challenge.  Scalars follow the reference's update order, so
the outputs are the same group elements / field elements the reference prover produces (tests/test_prover_kernels_gpu.py
replays inputs recorded from the reference prover and compares every L / R point and final scalar).
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

from .msm_accumulator import batch_fold, batch_mul, batch_mul_same_scalar, compute_MSM, compute_MSM_batch
from .py_arkworks_bls12381 import G1Point, Scalar
from .util import random_scalar

N_BLINDERS = 4                                                          # curdleproofs.py:24


def _inner(a: Sequence[Scalar], b: Sequence[Scalar]) -> Scalar:          # util.py:85-87
    acc = 0
    for x, y in zip(a, b):
        acc += x._v * y._v
    return Scalar(acc)


def ipa_rounds(crs_G_vec: Sequence[G1Point], crs_G_prime_vec: Sequence[G1Point], H: G1Point, vec_c: Sequence[Scalar],
               vec_d: Sequence[Scalar], next_gamma: Callable[[G1Point, G1Point, G1Point, G1Point], Scalar]):
    """ipa.py:117-151.  `H` is crs_H * beta (ipa.py:110); vec_c / vec_d are the blinded vectors (after ipa.py:107-109).
    -> (vec_L_C, vec_R_C, vec_L_D, vec_R_D, c_final, d_final)."""
    G, Gp, c, d = list(crs_G_vec), list(crs_G_prime_vec), list(vec_c), list(vec_d)
    n = len(c)
    assert n == len(d) == len(G) == len(Gp) and n & (n - 1) == 0
    LC, RC, LD, RD = [], [], [], []
    while n > 1:
        n //= 2
        c_L, c_R, d_L, d_R = c[:n], c[n:], d[:n], d[n:]
        G_L, G_R, Gp_L, Gp_R = G[:n], G[n:], Gp[:n], Gp[n:]
        # L_C = MSM(G_R, c_L) + H <c_L, d_R>;  L_D = MSM(G'_L, d_R);  R_C = MSM(G_L, c_R) + H <c_R, d_L>;  R_D = MSM(G'_R, d_L)
        L_C, L_D, R_C, R_D = compute_MSM_batch([(G_R + [H], c_L + [_inner(c_L, d_R)]), (Gp_L, d_R),
                                                (G_L + [H], c_R + [_inner(c_R, d_L)]), (Gp_R, d_L)])
        LC.append(L_C); RC.append(R_C); LD.append(L_D); RD.append(R_D)
        gamma = next_gamma(L_C, L_D, R_C, R_D)
        gamma_inv = gamma.inverse()
        c = [l + gamma_inv * r for l, r in zip(c_L, c_R)]
        d = [l + gamma * r for l, r in zip(d_L, d_R)]
        G = batch_fold(G_L, G_R, gamma)                                  # G_L[i] + G_R[i] * gamma
        Gp = batch_fold(Gp_L, Gp_R, gamma_inv)
    return LC, RC, LD, RD, c[0], d[0]


def same_msm_rounds(crs_G_vec: Sequence[G1Point], vec_T: Sequence[G1Point], vec_U: Sequence[G1Point], vec_x: Sequence[Scalar],
                    next_gamma: Callable[[G1Point, G1Point, G1Point, G1Point, G1Point, G1Point], Scalar]):
    """same_msm.py:93-130 (vec_x already blinded, :89-91).  -> (vec_L_A, vec_L_T, vec_L_U, vec_R_A, vec_R_T, vec_R_U, x_final)."""
    G, T, U, x = list(crs_G_vec), list(vec_T), list(vec_U), list(vec_x)
    n = len(x)
    assert n == len(G) == len(T) == len(U) and n & (n - 1) == 0
    out = [[] for _ in range(6)]
    while n > 1:
        n //= 2
        x_L, x_R = x[:n], x[n:]
        T_L, T_R, U_L, U_R, G_L, G_R = T[:n], T[n:], U[:n], U[n:], G[:n], G[n:]
        rnd = compute_MSM_batch([(G_R, x_L), (T_R, x_L), (U_R, x_L), (G_L, x_R), (T_L, x_R), (U_L, x_R)])    # L_A L_T L_U R_A R_T R_U
        for lst, p in zip(out, rnd):
            lst.append(p)
        gamma = next_gamma(*rnd)
        gamma_inv = gamma.inverse()
        x = [l + gamma_inv * r for l, r in zip(x_L, x_R)]
        folded = batch_fold(T_L + U_L + G_L, T_R + U_R + G_R, gamma)     # the three folds share gamma: one launch
        T, U, G = folded[:n], folded[n: 2 * n], folded[2 * n:]
    return (*out, x[0])


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
