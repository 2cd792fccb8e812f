"""Prover-side hot loops of the shuffle argument, re-cut for the GPU.

The reference prover spends its time in four patterns, all written as Python loops over single `G1Point` operators:

  * the halving rounds of the inner-product argument       ipa.py:117-151      4 MSMs of h terms + 2 folds of h points per round
  * the halving rounds of the same-MSM argument            same_msm.py:93-130  6 MSMs of h terms + 3 folds of h points per round
  * `shuffle_permute_and_commit_input`                     curdleproofs.py:301-321   2 ell same-scalar multiplications + 2 MSMs
  * the grand-product base change G'_i = G_i * beta^-(i+1) grand_prod.py:64-71 ell + 4 per-index multiplications

Here every halving round is ONE batched GPU MSM call (`compute_MSM_batch`: a handful of small MSMs ride one k_msm_small launch) and
NO fold of the bases at all -- the round challenges are folded into the scalars instead (see ipa_rounds_many) -- and the map patterns
are one `batch_mul*` launch each.  The `*_many` forms run several independent provers in step: round k of ALL of them is still one
MSM call (cross-proof batching; tools/gpu_prover_bench.py measures both).  The Fiat-Shamir
transcript stays with the caller, exactly where the reference has it: the round functions take a `next_gamma` callback that
receives the round's commitments (to absorb them) and returns the challenge.  Scalars follow the reference's update order, so
the outputs are the same group elements / field elements the reference prover produces (tests/test_prover_kernels_gpu.py
replays inputs recorded from the reference prover and compares every L / R point and final scalar).
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

from .msm_accumulator import batch_mul, batch_mul_same_scalar, compute_MSM, compute_MSM_batch
from .py_arkworks_bls12381 import CURVE_ORDER, G1Point, Scalar
from .util import random_scalar

N_BLINDERS = 4                                                          # curdleproofs.py:24


def _inner(a: Sequence[Scalar], b: Sequence[Scalar]) -> Scalar:          # util.py:85-87
    acc = 0
    for x, y in zip(a, b):
        acc += x._v * y._v
    return Scalar(acc)


def _halves(n0: int, n: int):
    """Original indices whose CURRENT position (j mod 2n, after the halvings so far) lies in the left / right half of a vector of length 2n."""
    left = [j for j in range(n0) if j % (2 * n) < n]
    right = [j for j in range(n0) if j % (2 * n) >= n]
    return left, right


def ipa_rounds_many(provers: Sequence[Tuple[Sequence[G1Point], Sequence[G1Point], G1Point, Sequence[Scalar], Sequence[Scalar]]],
                    next_gammas: Sequence[Callable[[G1Point, G1Point, G1Point, G1Point], Scalar]]):
    """ipa.py:117-151 for SEVERAL independent provers in step (same vector length): round k of all of them is ONE batched MSM call
    (4 MSMs per prover; up to 16 MSMs ride one k_msm_small launch).  provers[p] = (crs_G_vec, crs_G_prime_vec, H, vec_c, vec_d) with
    `H` = crs_H * beta (ipa.py:110) and the vectors already blinded (ipa.py:107-109).
    -> per prover (vec_L_C, vec_R_C, vec_L_D, vec_R_D, c_final, d_final).

    The reference folds the BASES every round (G = G_L + gamma G_R, ipa.py:142-146: h scalar multiplications, a 255-doubling chain
    each).  The prover never outputs a folded base -- only MSMs over them -- and a folded base is a fixed combination of the original
    ones, G^(k)_i = sum over j = i (mod n_k) of coef_j G_j with coef_j the product of the challenges of the rounds in which j sat in
    the right half.  So the bases stay what they were (normal forms cached, or resident on the device) and the challenges fold into
    the SCALARS: MSM(G^(k)_R, c_L) = sum over j in the right half of c_L[j mod 2n - n] coef_j G_j.  Same group elements, no fold
    launches at all (they were 15.8 of the 21 ms these rounds took: profiles/r03_prover_flows_v2.txt)."""
    st = [dict(G=list(pr[0]), Gp=list(pr[1]), H=pr[2], c=[x._v for x in pr[3]], d=[x._v for x in pr[4]], LC=[], RC=[], LD=[], RD=[],
               kGp0=(list(pr[5]) if len(pr) > 5 and pr[5] is not None else None)) for pr in provers]
    n0 = n = len(st[0]["c"])
    assert all(len(s["c"]) == len(s["d"]) == len(s["G"]) == len(s["Gp"]) == n for s in st) and n & (n - 1) == 0
    for s in st:
        # coef_j of G_j / G'_j.  A sixth tuple element gives the G' vector implicitly: G'_j = Gp[j] * coeffs[j] -- the grand-product
        # argument's base change G'_i = G_i beta^-(i+1) (grand_prod.py:64-71) then costs no scalar multiplication at all: pass the
        # CRS points themselves as Gp and the powers as coefficients
        s["kG"], s["kGp"] = [1] * n0, ([1] * n0 if s["kGp0"] is None else [v._v for v in s["kGp0"]])
        assert len(s["kGp"]) == n0
    R = CURVE_ORDER
    S = Scalar._raw
    while n > 1:
        n //= 2
        left, right = _halves(n0, n)
        jobs = []
        for s in st:
            c, d, kG, kGp, G, Gp = s["c"], s["d"], s["kG"], s["kGp"], s["G"], s["Gp"]
            ip_l = sum(c[i] * d[n + i] for i in range(n)) % R            # <c_L, d_R>
            ip_r = sum(c[n + i] * d[i] for i in range(n)) % R            # <c_R, d_L>
            # L_C = MSM(G_R, c_L) + H <c_L, d_R>;  L_D = MSM(G'_L, d_R);  R_C = MSM(G_L, c_R) + H <c_R, d_L>;  R_D = MSM(G'_R, d_L)
            jobs += [([G[j] for j in right] + [s["H"]], [S(c[j % (2 * n) - n] * kG[j] % R) for j in right] + [S(ip_l)]),
                     ([Gp[j] for j in left], [S(d[n + j % (2 * n)] * kGp[j] % R) for j in left]),
                     ([G[j] for j in left] + [s["H"]], [S(c[n + j % (2 * n)] * kG[j] % R) for j in left] + [S(ip_r)]),
                     ([Gp[j] for j in right], [S(d[j % (2 * n) - n] * kGp[j] % R) for j in right])]
        res = compute_MSM_batch(jobs)
        for i, (s, ng) in enumerate(zip(st, next_gammas)):
            L_C, L_D, R_C, R_D = res[4 * i: 4 * i + 4]
            s["LC"].append(L_C); s["RC"].append(R_C); s["LD"].append(L_D); s["RD"].append(R_D)
            gamma = ng(L_C, L_D, R_C, R_D)._v
            gamma_inv = pow(gamma, -1, R)
            c, d, kG, kGp = s["c"], s["d"], s["kG"], s["kGp"]
            s["c"] = [(c[i] + gamma_inv * c[n + i]) % R for i in range(n)]
            s["d"] = [(d[i] + gamma * d[n + i]) % R for i in range(n)]
            for j in right:                                              # G_L[i] + G_R[i] * gamma,  G'_L[i] + G'_R[i] * gamma^-1
                kG[j] = kG[j] * gamma % R
                kGp[j] = kGp[j] * gamma_inv % R
    return [(s["LC"], s["RC"], s["LD"], s["RD"], S(s["c"][0]), S(s["d"][0])) for s in st]


def ipa_rounds(crs_G_vec: Sequence[G1Point], crs_G_prime_vec: Sequence[G1Point], H: G1Point, vec_c: Sequence[Scalar],
               vec_d: Sequence[Scalar], next_gamma: Callable[[G1Point, G1Point, G1Point, G1Point], Scalar],
               G_prime_coeffs: Sequence[Scalar] = None):
    """ipa.py:117-151.  `H` is crs_H * beta (ipa.py:110); vec_c / vec_d are the blinded vectors (after ipa.py:107-109).
    G_prime_coeffs (optional): crs_G_prime_vec[j] stands for crs_G_prime_vec[j] * G_prime_coeffs[j] (see ipa_rounds_many).
    -> (vec_L_C, vec_R_C, vec_L_D, vec_R_D, c_final, d_final)."""
    return ipa_rounds_many([(crs_G_vec, crs_G_prime_vec, H, vec_c, vec_d, G_prime_coeffs)], [next_gamma])[0]


def same_msm_rounds_many(provers: Sequence[Tuple[Sequence[G1Point], Sequence[G1Point], Sequence[G1Point], Sequence[Scalar]]],
                         next_gammas: Sequence[Callable[..., Scalar]]):
    """same_msm.py:93-130 for several independent provers in step: per round ONE batched MSM call (6 MSMs per prover), the
    challenges folded into the scalars as in ipa_rounds_many (the three base vectors of a prover share one coefficient vector: all
    three fold with gamma, same_msm.py:122-126).  provers[p] = (crs_G_vec, vec_T, vec_U, vec_x) with vec_x already blinded (:89-91).
    -> per prover (vec_L_A, vec_L_T, vec_L_U, vec_R_A, vec_R_T, vec_R_U, x_final)."""
    st = [dict(G=list(G), T=list(T), U=list(U), x=[v._v for v in x], out=[[] for _ in range(6)]) for G, T, U, x in provers]
    n0 = n = len(st[0]["x"])
    assert all(len(s["x"]) == len(s["G"]) == len(s["T"]) == len(s["U"]) == n for s in st) and n & (n - 1) == 0
    for s in st:
        s["k"] = [1] * n0
    R = CURVE_ORDER
    S = Scalar._raw
    while n > 1:
        n //= 2
        left, right = _halves(n0, n)
        jobs = []
        for s in st:
            x, k = s["x"], s["k"]
            sc_l = [S(x[j % (2 * n) - n] * k[j] % R) for j in right]     # x_L against the right halves
            sc_r = [S(x[n + j % (2 * n)] * k[j] % R) for j in left]      # x_R against the left halves
            for V in (s["G"], s["T"], s["U"]):                            # L_A L_T L_U
                jobs.append(([V[j] for j in right], sc_l))
            for V in (s["G"], s["T"], s["U"]):                            # R_A R_T R_U
                jobs.append(([V[j] for j in left], sc_r))
        res = compute_MSM_batch(jobs)
        for i, (s, ng) in enumerate(zip(st, next_gammas)):
            rnd = res[6 * i: 6 * i + 6]
            for lst, pnt in zip(s["out"], rnd):
                lst.append(pnt)
            gamma = ng(*rnd)._v
            gamma_inv = pow(gamma, -1, R)
            x, k = s["x"], s["k"]
            s["x"] = [(x[i] + gamma_inv * x[n + i]) % R for i in range(n)]
            for j in right:
                k[j] = k[j] * gamma % R
    return [(*s["out"], S(s["x"][0])) for s in st]


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


def grand_product_coeffs(ell: int, n_blinders: int, beta_inv: Scalar) -> List[Scalar]:
    """The scalars of grand_prod.py:64-71 -- beta^-(i+1) for the ell G's, beta^-(ell+1) for the blinders -- for callers that hand the base
    change to ipa_rounds as `G_prime_coeffs` instead of materialising G' / H' (grand_product_bases below: one 2.4 ms launch)."""
    out, p = [], beta_inv
    for _ in range(ell):
        out.append(p)
        p = p * beta_inv
    return out + [p] * n_blinders


def grand_product_bases(crs_G_vec: Sequence[G1Point], crs_H_vec: Sequence[G1Point], beta_inv: Scalar) -> Tuple[List[G1Point], List[G1Point]]:
    """grand_prod.py:64-71: G'_i = G_i * beta^-(i+1), H'_i = H_i * beta^-(ell+1), one per-index launch."""
    ell = len(crs_G_vec)
    pows, p = [], beta_inv
    for _ in range(ell):
        pows.append(p)
        p = p * beta_inv
    res = batch_mul(list(crs_G_vec) + list(crs_H_vec), pows + [p] * len(crs_H_vec))
    return res[:ell], res[ell:]
