// Quad-lane EC operations for the latency-bound kernels (device only).
//
// One lane needs ~14.5 dependent field multiplications for an XYZZ addition and ~8 for a doubling, and a lone
// wave issues one half-rate VALU instruction per ~8 cycles whatever else the chip is doing, so a chain of EC
// operations in one lane costs ~25-35 us per link.  Here the 4 lanes of a DPP quad hold IDENTICAL copies of the
// operands and split the independent multiplications of each stage among themselves:
//     add :  [U1 U2 S1 S2] -> [P^2 R^2 ZZ1ZZ2 ZZZ1ZZZ2] -> [P^3  U1 P^2  (ZZ1ZZ2)P^2  -] -> [-  Y3  -  ZZZ3]      4.5 mul-times
//     dbl :  [U^2 X^2 - -] -> [U V  X V  (3X^2)^2  V ZZ] -> [Y3  W ZZZ  - -]                                        3.5 mul-times
// Every lane runs the SAME instruction stream (operands are picked with per-lane selects, results are broadcast
// with v_mov_dpp quad_perm), so there is no divergence inside a wave; all 4 lanes end with the same full result.
// Exceptional cases (identity operands, P + P, P - P) are decided from data all 4 lanes share, hence uniformly
// per quad, and fall back to the scalar formulas.
//
// Contract: all 4 lanes of a quad (lane & ~3 .. | 3) are active and pass identical arguments.
#pragma once
#include "g1_xyzz.h"

namespace cg1 {

template <int SRC>
__device__ __forceinline__ fp quad_bcast(const fp& v) {        // v as held by lane SRC of my quad
  fp r;
#pragma unroll
  for (int i = 0; i < NL; ++i)
    r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], SRC * 0x55, 0xf, 0xf, false);   // quad_perm:[SRC,SRC,SRC,SRC]
  return r;
}

__device__ __forceinline__ fp quad_sel(const fp& a0, const fp& a1, const fp& a2, const fp& a3, uint32_t q) {
  fp r;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    uint32_t lo = (q & 1u) ? a1.l[i] : a0.l[i];
    uint32_t hi = (q & 1u) ? a3.l[i] : a2.l[i];
    r.l[i] = (q & 2u) ? hi : lo;
  }
  return r;
}

// 2 * a
__device__ __forceinline__ xyzz quad_dbl(const xyzz& a, uint32_t q) {
  if (a.inf) return a;
  const fp U = fp_dbl(a.Y);
  // stage 1: q0 (and q2,q3 redundantly): V = U^2;  q1: XX = X^2
  const fp s1 = quad_sel(U, a.X, U, U, q);
  const fp m1 = fp_sqr(s1);
  const fp V = quad_bcast<0>(m1), XX = quad_bcast<1>(m1);
  const fp M = fp_add(fp_dbl(XX), XX);
  // stage 2: q0: W = U V;  q1: S = X V;  q2: MM = M M;  q3: ZZ3 = V ZZ
  const fp a2 = quad_sel(U, a.X, M, V, q), b2 = quad_sel(V, V, M, a.ZZ, q);
  const fp m2 = fp_mul(a2, b2);
  const fp W = quad_bcast<0>(m2), S = quad_bcast<1>(m2), MM = quad_bcast<2>(m2), ZZ3 = quad_bcast<3>(m2);
  const fp X3 = fp_norm(fp_add(MM, fp_dbl(fp_neg<3>(S))));
  // stage 3: q0 (and q2,q3): Y3 = M (S - X3) - W Y;  q1: ZZZ3 = W ZZZ (+ 0)
  const fp zero = fp_zero();
  const fp sx = fp_sub<12>(S, X3), ny = fp_neg<6>(a.Y);
  const fp a3 = quad_sel(M, W, M, M, q), b3 = quad_sel(sx, a.ZZZ, sx, sx, q);
  const fp c3 = quad_sel(W, zero, W, W, q), d3 = quad_sel(ny, zero, ny, ny, q);
  const fp m3 = fp_mul2(a3, b3, c3, d3);
  xyzz r;
  r.X = X3; r.Y = quad_bcast<0>(m3); r.ZZ = ZZ3; r.ZZZ = quad_bcast<1>(m3); r.inf = 0;
  return r;
}

// a + b
__device__ __forceinline__ xyzz quad_add(const xyzz& a, const xyzz& b, uint32_t q) {
  if (a.inf) return b;
  if (b.inf) return a;
  // stage 1: q0: U1 = X1 ZZ2;  q1: U2 = X2 ZZ1;  q2: S1 = Y1 ZZZ2;  q3: S2 = Y2 ZZZ1
  const fp a1 = quad_sel(a.X, b.X, a.Y, b.Y, q), b1 = quad_sel(b.ZZ, a.ZZ, b.ZZZ, a.ZZZ, q);
  const fp m1 = fp_mul(a1, b1);
  const fp U1 = quad_bcast<0>(m1), U2 = quad_bcast<1>(m1), S1 = quad_bcast<2>(m1), S2 = quad_bcast<3>(m1);
  const fp P = fp_sub<3>(U2, U1), R = fp_sub<3>(S2, S1);
  if (fp_is_zero_mod_p(P, 6)) {                         // same decision in all 4 lanes (identical data)
    if (fp_is_zero_mod_p(R, 6)) return xyzz_dbl(a);
    return xyzz_identity();
  }
  // stage 2: q0: PP = P P;  q1: RR = R R;  q2: ZZ12 = ZZ1 ZZ2;  q3: ZZZ12 = ZZZ1 ZZZ2
  const fp a2 = quad_sel(P, R, a.ZZ, a.ZZZ, q), b2 = quad_sel(P, R, b.ZZ, b.ZZZ, q);
  const fp m2 = fp_mul(a2, b2);
  const fp PP = quad_bcast<0>(m2), RR = quad_bcast<1>(m2), ZZ12 = quad_bcast<2>(m2), ZZZ12 = quad_bcast<3>(m2);
  // stage 3: q0 (and q3): PPP = P PP;  q1: Q = U1 PP;  q2: ZZ3 = ZZ12 PP
  const fp a3 = quad_sel(P, U1, ZZ12, P, q);
  const fp m3 = fp_mul(a3, PP);
  const fp PPP = quad_bcast<0>(m3), Q = quad_bcast<1>(m3), ZZ3 = quad_bcast<2>(m3);
  const fp X3 = fp_norm(fp_add(fp_add(RR, fp_neg<3>(PPP)), fp_dbl(fp_neg<3>(Q))));
  // stage 4: q0,q1,q2: Y3 = R (Q - X3) - S1 PPP;  q3: ZZZ3 = ZZZ12 PPP (+ 0)
  const fp zero = fp_zero();
  const fp qx = fp_sub<12>(Q, X3), ns1 = fp_neg<3>(S1);
  const fp a4 = quad_sel(R, R, R, ZZZ12, q), b4 = quad_sel(qx, qx, qx, PPP, q);
  const fp c4 = quad_sel(PPP, PPP, PPP, zero, q), d4 = quad_sel(ns1, ns1, ns1, zero, q);
  const fp m4 = fp_mul2(a4, b4, c4, d4);
  xyzz r;
  r.X = X3; r.Y = quad_bcast<0>(m4); r.ZZ = ZZ3; r.ZZZ = quad_bcast<3>(m4); r.inf = 0;
  return r;
}

}  // namespace cg1
