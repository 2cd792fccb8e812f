// Verifier-side Fr vector work on the device: the scalar rows of the shuffle verifier's MSM statement.
// Part of the single translation unit csrc/msm_gpu.hip (included at global scope; uses csrc/fr.h compiled for the device).
//
// Stands behind the scalar algebra of the reference verifier -- verification scalars s_i = prod gamma_j^{bit j of i}
// (ipa.py:155-186, same_msm.py:146-182, util.py:71-78), their inverses, u_i = beta^-(i+1) (grand_prod.py:64-71), c*s, d*s^-1*u
// (ipa.py:216,227-229), x*s (same_msm.py:213), the accumulator's rho-weighted scalar merge (msm_accumulator.py:43-58) -- which
// csrc/shuffle_verify.cpp's front-end expands into one row of 4 ell + 19 + 10 lg scalars over the proof's own points and one row
// of ell + 9 scalars over the CRS points.  The host keeps the transcript (a serial sponge) and hands over, per proof, the
// challenges it drew plus a handful of derived scalars (5.7 KB instead of 23 KB of rows); one 128-thread block per proof
// computes every row entry in closed form, thread i owning vector index i:
//
//   own points:  R_i: -w7 a_i   S_i: -w8 a_i   T_i: -x5 sm_i   U_i: -x6 sm_i            (w7 = rho6, w8 = rho7, x5 = rho4 x, x6 = rho5 x)
//   CRS:         G_i: -rho0 beta_p - rho1 c s_i - rho2 d s_inv_i u_i - rho3 x sm_i      (i < ell; blinder slots without the first / last term)
//   plus ~50 single entries (L / R points of the three halving arguments, commitments, H, G_t, G_u, G_sum, H_sum).
//
// The rows are byte-identical to the host front-end's (tests/test_shuffle_rows_gpu.py).  A second kernel adds the CRS rows of
// the batch's live proofs (cg1_shuffle_sum_crs_scalars) behind the own-point scalars, where the merged MSM expects them.
#pragma once
#include "fr.h"

namespace cg1rows {
using cg1fr::fr;

// row-input block per proof (32-byte little-endian canonical scalars), written by cg1_shuffle_prepare_inputs
struct RowIn {
  size_t ell, lg;
  __host__ __device__ size_t head() const { return 0; }                 // alpha_p beta_p alpha_g beta_g alpha_i beta_i alpha_s alpha_m
  __host__ __device__ size_t gam() const { return 8; }
  __host__ __device__ size_t gm() const { return 8 + lg; }
  __host__ __device__ size_t a() const { return 8 + 2 * lg; }
  __host__ __device__ size_t beta_inv() const { return 8 + 2 * lg + ell; }
  __host__ __device__ size_t inner_prod() const { return beta_inv() + 1; }
  __host__ __device__ size_t gam_inv() const { return beta_inv() + 2; }
  __host__ __device__ size_t gm_inv() const { return gam_inv() + lg; }
  __host__ __device__ size_t fields() const { return gm_inv() + lg; }    // c_fin d_fin z_k z_t z_u x_fin
  __host__ __device__ size_t rho() const { return fields() + 6; }
  __host__ __device__ size_t count() const { return rho() + 12; }
};

// positions of a proof's own points and of the CRS points (the same layout as csrc/shuffle_verify.cpp `Layout`)
struct Lay {
  size_t ell, lg;
  __device__ size_t base() const { return 4 * ell; }
  __device__ size_t cmA1() const { return base() + 12 + 4 * lg; }
  __device__ size_t count() const { return 4 * ell + 19 + 10 * lg; }
  __device__ size_t ncrs() const { return ell + 9; }
};

// out-of-line Fr product for this kernel: ~60 call sites of an ~800-instruction body would make a 560 KB kernel
__device__ __noinline__ fr mulx(const fr& a, const fr& b) { return cg1fr::fr_mul(a, b); }

__device__ __noinline__ fr ld(const uint8_t* blk, size_t idx) {
  fr v;
  cg1fr::fr_from_le32(blk + 32 * idx, v);              // canonical by construction (written by the host front-end)
  return v;
}
__device__ __noinline__ void st(uint8_t* dst, size_t idx, const fr& v) { cg1fr::fr_to_le32(v, dst + 32 * idx); }
__device__ inline fr powx(fr base, uint64_t e) {        // fr_pow_u64 over the out-of-line product
  fr acc = cg1fr::fr_one();
  while (e) {
    if (e & 1) acc = mulx(acc, base);
    base = mulx(base, base);
    e >>= 1;
  }
  return acc;
}

// prod over j of g_j^{bit j of i, MSB first}  (fold_scalars of the host front-end, element i)
__device__ inline fr fold_elem(const uint8_t* blk, size_t first, size_t lg, uint32_t i) {
  fr acc = cg1fr::fr_one();
  for (size_t j = 0; j < lg; ++j)
    if ((i >> (lg - 1 - j)) & 1u) acc = mulx(acc, ld(blk, first + j));
  return acc;
}

__global__ void __launch_bounds__(128) k_shuffle_rows(const uint8_t* __restrict__ rowin, const int32_t* __restrict__ host_status,
                                                      const uint8_t* __restrict__ point_status, uint32_t ell32, uint32_t lg32,
                                                      uint8_t* __restrict__ out_scalars, uint8_t* __restrict__ out_crs_rows,
                                                      int32_t* __restrict__ status_out) {
  using namespace cg1fr;
  const size_t ell = ell32, lg = lg32, n = ell + 4;
  const RowIn R{ell, lg};
  const Lay L{ell, lg};
  const size_t proof = blockIdx.x;
  const uint8_t* blk = rowin + proof * R.count() * 32;
  uint8_t* sc = out_scalars + proof * L.count() * 32;
  uint8_t* cs = out_crs_rows + proof * L.ncrs() * 32;
  // a proof with an undecodable own point is rejected (BufReader.read_g1 raises, util.py:143-147)
  int bad = 0;
  for (size_t k = threadIdx.x; k < L.count(); k += blockDim.x) bad |= point_status[proof * L.count() + k] != 0;
  bad = __syncthreads_or(bad);
  int status = host_status[proof];
  if (!status && bad) status = 2;                      // CG1_SHUFFLE_BAD_POINT
  if (threadIdx.x == 0) status_out[proof] = status;
  if (status) {                                        // a rejected proof contributes nothing to the merged check
    uint32_t* z = reinterpret_cast<uint32_t*>(sc);
    for (size_t k = threadIdx.x; k < L.count() * 8; k += blockDim.x) z[k] = 0;
    z = reinterpret_cast<uint32_t*>(cs);
    for (size_t k = threadIdx.x; k < L.ncrs() * 8; k += blockDim.x) z[k] = 0;
    return;
  }
  const fr rho0 = ld(blk, R.rho() + 0), rho1 = ld(blk, R.rho() + 1), rho2 = ld(blk, R.rho() + 2), rho3 = ld(blk, R.rho() + 3),
           rho4 = ld(blk, R.rho() + 4), rho5 = ld(blk, R.rho() + 5);
  const fr c_fin = ld(blk, R.fields() + 0), d_fin = ld(blk, R.fields() + 1), x_fin = ld(blk, R.fields() + 5);
  const fr beta_inv = ld(blk, R.beta_inv());
  const fr wc = mulx(rho1, c_fin), wd = mulx(rho2, d_fin);
  const fr x4 = mulx(rho3, x_fin), x5 = mulx(rho4, x_fin), x6 = mulx(rho5, x_fin);
  // ---- vector entries: thread i owns index i of the n-vectors
  for (size_t i = threadIdx.x; i < n; i += blockDim.x) {
    const fr s = fold_elem(blk, R.gam(), lg, (uint32_t)i), s_inv = fold_elem(blk, R.gam_inv(), lg, (uint32_t)i);
    const fr sm = fold_elem(blk, R.gm(), lg, (uint32_t)i);
    const fr u = powx(beta_inv, (i < ell ? i : ell) + 1);          // beta^-(i+1); beta^-(ell+1) for the blinder slots
    fr g = fr_neg(fr_add(mulx(wc, s), mulx(wd, mulx(s_inv, u))));  // CRS slot i (vec_G | vec_H): E2 and E3
    if (i < ell) {
      g = fr_sub(g, fr_add(mulx(rho0, ld(blk, R.head() + 1)), mulx(x4, sm)));     // E1: -rho0 beta_p;  same-MSM: -x4 sm_i
      const fr a = ld(blk, R.a() + i);
      st(sc, i, fr_neg(mulx(ld(blk, R.rho() + 6), a)));                 // R_i
      st(sc, ell + i, fr_neg(mulx(ld(blk, R.rho() + 7), a)));           // S_i
      st(sc, 2 * ell + i, fr_neg(mulx(x5, sm)));                        // T_i
      st(sc, 3 * ell + i, fr_neg(mulx(x6, sm)));                        // U_i
    } else if (i < ell + 2) {
      g = fr_sub(g, mulx(x4, sm));                                      // vec_H[0], vec_H[1] stand in G'' (curdleproofs.py:206-224)
    }
    st(cs, i, g);
  }
  // ---- single entries: thread j < lg takes the ten L / R points of halving round j, a few more threads the rest
  const size_t t = threadIdx.x;
  if (t < lg) {
    const fr g = ld(blk, R.gam() + t), gi = ld(blk, R.gam_inv() + t), m = ld(blk, R.gm() + t), mi = ld(blk, R.gm_inv() + t);
    const size_t b = L.base() + 12;
    st(sc, b + t, mulx(rho1, g));                   // L_C[j]
    st(sc, b + lg + t, mulx(rho1, gi));             // R_C[j]
    st(sc, b + 2 * lg + t, mulx(rho2, g));          // L_D[j]
    st(sc, b + 3 * lg + t, mulx(rho2, gi));         // R_D[j]
    const size_t q = L.cmA1() + 7;
    st(sc, q + t, mulx(rho3, m));                   // L_A[j]
    st(sc, q + lg + t, mulx(rho4, m));              // L_T[j]
    st(sc, q + 2 * lg + t, mulx(rho5, m));          // L_U[j]
    st(sc, q + 3 * lg + t, mulx(rho3, mi));         // R_A[j]
    st(sc, q + 4 * lg + t, mulx(rho4, mi));         // R_T[j]
    st(sc, q + 5 * lg + t, mulx(rho5, mi));         // R_U[j]
  } else if (t == 32) {
    const fr alpha_p = ld(blk, 0), alpha_i = ld(blk, 4), alpha_s = ld(blk, 6), alpha_m = ld(blk, 7);
    const fr w1 = ld(blk, R.rho() + 8), w2 = ld(blk, R.rho() + 9), w3 = ld(blk, R.rho() + 10), w4 = ld(blk, R.rho() + 11);
    const fr z_k = ld(blk, R.fields() + 2);
    const fr r3a = mulx(rho3, alpha_m);
    const size_t b = L.base();
    st(sc, b + 0, fr_neg(mulx(rho0, alpha_p)));                                  // M
    st(sc, b + 1, fr_sub(r3a, rho0));                                              // A
    st(sc, b + 2, fr_sub(r3a, mulx(w1, alpha_s)));                               // T_1
    st(sc, b + 3, fr_sub(mulx(rho4, alpha_m), mulx(w2, alpha_s)));             // T_2
    st(sc, b + 4, fr_sub(r3a, mulx(w3, alpha_s)));                               // U_1
    st(sc, b + 5, fr_sub(mulx(rho5, alpha_m), mulx(w4, alpha_s)));             // U_2
    st(sc, b + 6, fr_add(mulx(w2, z_k), ld(blk, R.rho() + 6)));                  // R
    st(sc, b + 7, fr_add(mulx(w4, z_k), ld(blk, R.rho() + 7)));                  // S
    st(sc, b + 8, fr_add(rho0, mulx(rho2, alpha_i)));                            // B
    st(sc, b + 9, mulx(rho1, alpha_i));                                          // C
    st(sc, b + 10, rho1);                                                          // B_c
    st(sc, b + 11, rho2);                                                          // B_d
    st(sc, L.cmA1() + 0, fr_neg(w1)); st(sc, L.cmA1() + 1, fr_neg(w2));            // cm_A
    st(sc, L.cmA1() + 2, fr_neg(w3)); st(sc, L.cmA1() + 3, fr_neg(w4));            // cm_B
    st(sc, L.cmA1() + 4, rho3); st(sc, L.cmA1() + 5, rho4); st(sc, L.cmA1() + 6, rho5);   // B_a B_t B_u
  } else if (t == 64) {
    const fr alpha_g = ld(blk, 2), alpha_i = ld(blk, 4), beta_i = ld(blk, 5);
    const fr w1 = ld(blk, R.rho() + 8), w2 = ld(blk, R.rho() + 9), w3 = ld(blk, R.rho() + 10), w4 = ld(blk, R.rho() + 11);
    const fr z_t = ld(blk, R.fields() + 3), z_u = ld(blk, R.fields() + 4);
    const fr sm2 = fold_elem(blk, R.gm(), lg, (uint32_t)(ell + 2)), sm3 = fold_elem(blk, R.gm(), lg, (uint32_t)(ell + 3));
    const fr hcoef = mulx(beta_i, fr_sub(mulx(mulx(alpha_i, alpha_i), ld(blk, R.inner_prod())), mulx(c_fin, d_fin)));
    const fr wa = mulx(rho2, alpha_i);
    // H:  E2's H coefficient, z_t and z_u of the same-scalar argument, the T' / U' blinder slots of the same-MSM argument
    fr h = fr_add(mulx(rho1, hcoef), fr_add(mulx(w2, z_t), mulx(w4, z_u)));
    h = fr_sub(h, fr_add(mulx(x5, sm2), mulx(x6, sm3)));
    st(cs, ell + 4, h);
    st(cs, ell + 5, fr_sub(mulx(w1, z_t), mulx(x4, sm2)));                     // G_t
    st(cs, ell + 6, fr_sub(mulx(w3, z_u), mulx(x4, sm3)));                     // G_u
    st(cs, ell + 7, fr_neg(mulx(wa, beta_inv)));                                 // G_sum
    st(cs, ell + 8, mulx(wa, alpha_g));                                          // H_sum
  }
}

// out[j] = sum over the proofs with status == 0 of crs_rows[i][j]  (cg1_shuffle_sum_crs_scalars on the device).
// One block per slot j: its 256 threads take the proofs t, t + 256, ..., then fold through LDS.  Rows are canonical
// little-endian integers: they add mod r as they are (and addition mod r of canonical values does not depend on the order).
__global__ void __launch_bounds__(256) k_crs_row_sum(const uint8_t* __restrict__ crs_rows, const int32_t* __restrict__ status,
                                                     uint32_t n_proofs, uint32_t ncrs, uint8_t* __restrict__ out) {
  __shared__ fr part[256];
  const uint32_t j = blockIdx.x, t = threadIdx.x;
  fr acc = cg1fr::fr_zero();
  for (uint32_t i = t; i < n_proofs; i += 256) {
    if (status[i]) continue;
    fr v;
    memcpy(v.l, crs_rows + ((size_t)i * ncrs + j) * 32, 32);
    acc = cg1fr::fr_add(acc, v);
  }
  part[t] = acc;
  __syncthreads();
  for (uint32_t d = 128; d >= 1; d >>= 1) {
    if (t < d) part[t] = cg1fr::fr_add(part[t], part[t + d]);
    __syncthreads();
  }
  if (t == 0) memcpy(out + 32 * (size_t)j, part[0].l, 32);
}

}  // namespace cg1rows
