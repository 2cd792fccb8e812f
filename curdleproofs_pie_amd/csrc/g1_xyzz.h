// G1 group law on y^2 = x^3 + 4 in extended Jacobian ("XYZZ") coordinates over fp28.h.
//   x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2.   Mixed add 8M+2S, affine+affine 4M+2S, full add 12M+2S, double 6M+4S (a = 0).
// Formulas: the standard madd-2008-s / add-2008-s / dbl-2008-s-1 of the EFD, with every exceptional
// case handled exactly (P+P, P+(-P), identity operands) -- the callers of compute_MSM really do pass
// duplicate bases and Z1 (curdleproofs.py:124-136 in the reference), and results must be bit-exact.
//
// Coordinate invariant for a stored xyzz value (checked in the CG1_CHECK_BOUNDS build):
//   all four coordinates in N-limb form (limbs 0..12 < 2^28);  value(X) < 11p, value(Y) < 5p,
//   value(ZZ), value(ZZZ) < 2p;  `inf != 0` marks the identity (coordinates then meaningless).
// Affine inputs: x in N-form (< 2p); y either N-form or the lazy negation 3p - y (limbs < 2^29).
#pragma once
#include "fp28.h"

namespace cg1 {

struct xyzz { fp X, Y, ZZ, ZZZ; uint32_t inf; };
struct aff  { fp x, y; };

CG1_HD xyzz xyzz_identity() { xyzz r; r.X = fp_zero(); r.Y = fp_zero(); r.ZZ = fp_zero(); r.ZZZ = fp_zero(); r.inf = 1; return r; }

CG1_HD xyzz xyzz_from_affine(const fp& x, const fp& y_maybe_lazy) {
  xyzz r; r.X = x; r.Y = fp_norm(y_maybe_lazy); r.ZZ = fp_one(); r.ZZZ = fp_one(); r.inf = 0; return r;
}

// 2 * (X1, Y1, ZZ1, ZZZ1).  E(Fp) has odd order, so Y1 != 0 for every finite point.
CG1_HD xyzz xyzz_dbl(const xyzz& a) {
  if (a.inf) return a;
  xyzz r;
  fp U = fp_dbl(a.Y);                       // limbs < 2^29, value < 10p
  fp V = fp_sqr(U);
  fp W = fp_mul(U, V);
  fp S = fp_mul(a.X, V);
  fp XX = fp_sqr(a.X);
  fp M = fp_add(fp_dbl(XX), XX);            // limbs < 3*2^28
  fp MM = fp_sqr(M);
  fp X3 = fp_norm(fp_add(MM, fp_dbl(fp_neg<3>(S))));          // value < 7.1p
  r.X = X3;
  r.Y = fp_mul2(M, fp_sub<12>(S, X3), W, fp_neg<6>(a.Y));     // M(S-X3) - W*Y1 in one reduction; N-form, < 1.1p
  r.ZZ = fp_mul(V, a.ZZ);
  r.ZZZ = fp_mul(W, a.ZZZ);
  r.inf = 0;
  return r;
}

// acc + (x2, y2) with (x2, y2) a finite affine point.
CG1_HD xyzz xyzz_madd(const xyzz& a, const fp& x2, const fp& y2) {
  if (a.inf) return xyzz_from_affine(x2, y2);
  fp U2 = fp_mul(x2, a.ZZ);
  fp S2 = fp_mul(y2, a.ZZZ);
  fp P = fp_sub<12>(U2, a.X);               // limbs < 2^28 + 2^29, value < 13.1p
  fp R = fp_sub<6>(S2, a.Y);                // value < 7.1p
  if (fp_is_zero_mod_p(P, 14)) {            // same x: doubling or cancellation (rare, divergent)
    if (fp_is_zero_mod_p(R, 8)) return xyzz_dbl(xyzz_from_affine(x2, y2));
    return xyzz_identity();
  }
  xyzz r;
  fp PP = fp_sqr(P);
  fp PPP = fp_mul(P, PP);
  fp Q = fp_mul(a.X, PP);
  fp RR = fp_sqr(R);
  fp X3 = fp_norm(fp_add(fp_add(RR, fp_neg<3>(PPP)), fp_dbl(fp_neg<3>(Q))));   // value < 10.1p
  r.X = X3;
  r.Y = fp_mul2(R, fp_sub<12>(Q, X3), PPP, fp_neg<6>(a.Y));   // R(Q-X3) - Y1*PPP in one reduction; N-form, < 1.1p
  r.ZZ = fp_mul(a.ZZ, PP);
  r.ZZZ = fp_mul(a.ZZZ, PPP);
  r.inf = 0;
  return r;
}

// The common case of xyzz_madd, updating `a` in place: false (and `a` untouched) when the addition is one of the exceptional
// cases (a is the identity, or the operands share their x coordinate) -- the caller then takes xyzz_madd.  Keeping the
// exceptional results out of the hot loop's control flow lets the accumulator stay in the same registers across iterations.
CG1_HD bool xyzz_madd_fast(xyzz& a, const fp& x2, const fp& y2) {
  if (a.inf) return false;
  fp U2 = fp_mul(x2, a.ZZ);
  fp S2 = fp_mul(y2, a.ZZZ);
  fp P = fp_sub<12>(U2, a.X);
  fp R = fp_sub<6>(S2, a.Y);
  if (fp_is_zero_mod_p(P, 14)) return false;
  fp PP = fp_sqr(P);
  fp PPP = fp_mul(P, PP);
  fp Q = fp_mul(a.X, PP);
  fp RR = fp_sqr(R);
  fp X3 = fp_norm(fp_add(fp_add(RR, fp_neg<3>(PPP)), fp_dbl(fp_neg<3>(Q))));
  fp Y3 = fp_mul2(R, fp_sub<12>(Q, X3), PPP, fp_neg<6>(a.Y));
  a.X = X3;
  a.Y = Y3;
  a.ZZ = fp_mul(a.ZZ, PP);
  a.ZZZ = fp_mul(a.ZZZ, PPP);
  return true;
}

// (x1, y1) + (x2, y2), both finite affine points (mmadd-2007-bl shape: 2M + 2S + one fused pair instead of madd's
// 6M + 2S + pair -- the first addition of every bucket chunk in k_accumulate has two affine operands).
CG1_HD xyzz xyzz_mmadd(const fp& x1, const fp& y1_maybe_lazy, const fp& x2, const fp& y2_maybe_lazy) {
  const fp y1 = fp_norm(y1_maybe_lazy), y2 = fp_norm(y2_maybe_lazy);    // N-limb form, value <= 3p
  fp P = fp_sub<3>(x2, x1);                 // value < 5p
  fp R = fp_sub<6>(y2, y1);                 // value < 9p
  if (fp_is_zero_mod_p(P, 6)) {             // same x: doubling or cancellation (rare, divergent)
    if (fp_is_zero_mod_p(R, 10)) return xyzz_dbl(xyzz_from_affine(x2, y2));
    return xyzz_identity();
  }
  xyzz r;
  fp PP = fp_sqr(P);
  fp PPP = fp_mul(P, PP);
  fp Q = fp_mul(x1, PP);
  fp RR = fp_sqr(R);
  fp X3 = fp_norm(fp_add(fp_add(RR, fp_neg<3>(PPP)), fp_dbl(fp_neg<3>(Q))));
  r.X = X3;
  r.Y = fp_mul2(R, fp_sub<12>(Q, X3), PPP, fp_neg<6>(y1));    // R(Q-X3) - y1*PPP in one reduction
  r.ZZ = PP;
  r.ZZZ = PPP;
  r.inf = 0;
  return r;
}

// a + b, both XYZZ.
CG1_HD xyzz xyzz_add(const xyzz& a, const xyzz& b) {
  if (a.inf) return b;
  if (b.inf) return a;
  fp U1 = fp_mul(a.X, b.ZZ);
  fp U2 = fp_mul(b.X, a.ZZ);
  fp S1 = fp_mul(a.Y, b.ZZZ);
  fp S2 = fp_mul(b.Y, a.ZZZ);
  fp P = fp_sub<3>(U2, U1);                 // value < 4.1p
  fp R = fp_sub<3>(S2, S1);
  if (fp_is_zero_mod_p(P, 6)) {
    if (fp_is_zero_mod_p(R, 6)) return xyzz_dbl(a);
    return xyzz_identity();
  }
  xyzz r;
  fp PP = fp_sqr(P);
  fp PPP = fp_mul(P, PP);
  fp Q = fp_mul(U1, PP);
  fp RR = fp_sqr(R);
  fp X3 = fp_norm(fp_add(fp_add(RR, fp_neg<3>(PPP)), fp_dbl(fp_neg<3>(Q))));
  r.X = X3;
  r.Y = fp_mul2(R, fp_sub<12>(Q, X3), PPP, fp_neg<3>(S1));    // R(Q-X3) - S1*PPP in one reduction
  r.ZZ = fp_mul(fp_mul(a.ZZ, b.ZZ), PP);
  r.ZZZ = fp_mul(fp_mul(a.ZZZ, b.ZZZ), PPP);
  r.inf = 0;
  return r;
}

// ---- canonical export: XYZZ with canonical coordinates in the HOST's Montgomery form (radix 2^384), 4 x 12 words
// (+ a flag word).  The host reinterprets the words as its field elements and maps the point to Jacobian
// (X*ZZ, Y*ZZZ, ZZ) without an inversion.
// ------------------------------------------------------------------ Jacobian coordinates, for long runs of doublings
// (X, Y, Z) with x = X / Z^2, y = Y / Z^3.  A doubling here is 3 products + 4 squarings (2 380 multiplier operations) against the
// 4 products + 3 squarings + one fused pair (3 059) of xyzz_dbl: the subgroup test of the checked decompression is 126 doublings and
// 12 additions per point, so it runs on these; the MSM's buckets stay on XYZZ, whose MIXED addition is the cheaper one.
// Invariants between operations: X carry-propagated (limbs 0..12 < 2^28) with value < 10.1p; Z in N-form; Y either N-form (after an
// addition) or lazily reduced with limbs < 2^28 + 2^29 and value < 13.1p (after a doubling) -- the additions first bring it back to
// N-form with one product by the Montgomery one.
struct jacp { fp X, Y, Z; uint32_t inf; };

CG1_HD jacp jacp_identity() { jacp r; r.X = fp_zero(); r.Y = fp_one(); r.Z = fp_zero(); r.inf = 1; return r; }
CG1_HD jacp jacp_from_affine(const fp& x, const fp& y_maybe_lazy) {
  jacp r; r.X = x; r.Y = fp_norm(y_maybe_lazy); r.Z = fp_one(); r.inf = 0; return r;
}

// 2 * (X1, Y1, Z1) on y^2 = x^3 + 4 (a = 0):  A = X1^2, B = Y1^2, C = B^2, D = 4 X1 B, E = 3A, X3 = E^2 - 2D, Y3 = E (D - X3) - 8C,
// Z3 = 2 Y1 Z1.  (D as a product rather than (X1 + B)^2 - A - C: the lazy subtractions of that form would need two carry passes.)
CG1_HD jacp jacp_dbl(const jacp& a) {
  if (a.inf) return a;
  jacp r;
  const fp A = fp_sqr(a.X);
  const fp B = fp_sqr(a.Y);                                     // limbs of Y < 2^30
  const fp C = fp_sqr(B);
  const fp D = fp_mul(a.X, fp_dbl(fp_dbl(B)));                  // 4B: limbs < 2^30, value < 4.4p  ->  N-form
  const fp E = fp_add(fp_dbl(A), A);                            // limbs < 3 * 2^28, value < 3.3p
  const fp F = fp_sqr(E);
  const fp X3 = fp_norm(fp_add(F, fp_dbl(fp_neg<3>(D))));       // value < 7.1p
  const fp C8 = fp_norm(fp_dbl(fp_dbl(fp_dbl(C))));             // value < 8.8p, limbs 0..12 < 2^28
  r.X = X3;
  r.Y = fp_add(fp_mul(E, fp_sub<12>(D, X3)), fp_neg<12>(C8));   // limbs < 2^28 + 2^29, value < 13.1p
  r.Z = fp_mul(fp_dbl(a.Y), a.Z);
  r.inf = 0;
  return r;
}

// a + (x2, y2), (x2, y2) a finite affine point: xyzz_madd's formulas with ZZ = Z1^2, ZZZ = Z1^3 and Z3 = Z1 * P.
CG1_HD jacp jacp_madd(const jacp& a, const fp& x2, const fp& y2) {
  if (a.inf) return jacp_from_affine(x2, y2);
  const fp Y1 = fp_mul(a.Y, fp_one());                          // back to N-form (a product by the Montgomery one keeps the value)
  const fp ZZ = fp_sqr(a.Z), ZZZ = fp_mul(ZZ, a.Z);
  const fp U2 = fp_mul(x2, ZZ);
  const fp S2 = fp_mul(y2, ZZZ);
  const fp P = fp_sub<12>(U2, a.X);                             // value < 13.1p
  const fp R = fp_sub<6>(S2, Y1);                               // value < 7.1p
  if (fp_is_zero_mod_p(P, 14)) {
    if (fp_is_zero_mod_p(R, 8)) return jacp_dbl(jacp_from_affine(x2, y2));
    return jacp_identity();
  }
  jacp r;
  const fp PP = fp_sqr(P);
  const fp PPP = fp_mul(P, PP);
  const fp Q = fp_mul(a.X, PP);
  const fp RR = fp_sqr(R);
  const fp X3 = fp_norm(fp_add(fp_add(RR, fp_neg<3>(PPP)), fp_dbl(fp_neg<3>(Q))));   // value < 10.1p
  r.X = X3;
  r.Y = fp_mul2(R, fp_sub<12>(Q, X3), PPP, fp_neg<6>(Y1));     // N-form
  r.Z = fp_mul(a.Z, P);
  r.inf = 0;
  return r;
}

// a + b, both Jacobian: xyzz_add's formulas, Z3 = Z1 Z2 P.
CG1_HD jacp jacp_add(const jacp& a, const jacp& b) {
  if (a.inf) return b;
  if (b.inf) return a;
  const fp Y1 = fp_mul(a.Y, fp_one()), Y2 = fp_mul(b.Y, fp_one());
  const fp ZZ1 = fp_sqr(a.Z), ZZ2 = fp_sqr(b.Z);
  const fp U1 = fp_mul(a.X, ZZ2);
  const fp U2 = fp_mul(b.X, ZZ1);
  const fp S1 = fp_mul(Y1, fp_mul(ZZ2, b.Z));
  const fp S2 = fp_mul(Y2, fp_mul(ZZ1, a.Z));
  const fp P = fp_sub<3>(U2, U1);
  const fp R = fp_sub<3>(S2, S1);
  if (fp_is_zero_mod_p(P, 6)) {
    if (fp_is_zero_mod_p(R, 6)) return jacp_dbl(a);
    return jacp_identity();
  }
  jacp r;
  const fp PP = fp_sqr(P);
  const fp PPP = fp_mul(P, PP);
  const fp Q = fp_mul(U1, PP);
  const fp RR = fp_sqr(R);
  const fp X3 = fp_norm(fp_add(fp_add(RR, fp_neg<3>(PPP)), fp_dbl(fp_neg<3>(Q))));
  r.X = X3;
  r.Y = fp_mul2(R, fp_sub<12>(Q, X3), PPP, fp_neg<3>(S1));
  r.Z = fp_mul(fp_mul(a.Z, b.Z), P);
  r.inf = 0;
  return r;
}

// P = (x, y) on the curve (Montgomery limbs, N-form) is in the prime-order subgroup G1  <=>  [z^2] P == phi(P) + P with
// phi(x, y) = (beta x, y)  (the endomorphism acts on G1 as multiplication by z^2 - 1; on no other point of E(Fp) does it).
// [z^2] P = [|z|] [|z|] P with |z| = 0xd201000000010000 (six set bits): 2 x (63 doublings + 5 additions) instead of the
// 128 doublings + ~22 additions of a plain ladder over z^2.
CG1_HD bool g1_in_subgroup(const fp& x, const fp& y) {
  constexpr uint64_t ZABS = 0xd201000000010000ull;
  constexpr uint32_t bt[NL] = {D_BETA[0], D_BETA[1], D_BETA[2], D_BETA[3], D_BETA[4], D_BETA[5], D_BETA[6], D_BETA[7], D_BETA[8], D_BETA[9], D_BETA[10], D_BETA[11], D_BETA[12], D_BETA[13]};
  fp beta; for (int k = 0; k < NL; ++k) beta.l[k] = bt[k];
  jacp q = jacp_from_affine(x, y);                 // top bit of |z|
#pragma unroll 1
  for (int bit = 62; bit >= 0; --bit) {
    q = jacp_dbl(q);
    if ((ZABS >> bit) & 1ull) q = jacp_madd(q, x, y);
  }
  jacp acc = q;
#pragma unroll 1
  for (int bit = 62; bit >= 0; --bit) {
    acc = jacp_dbl(acc);
    if ((ZABS >> bit) & 1ull) acc = jacp_add(acc, q);
  }
  const fp yneg = fp_neg<3>(y);
  acc = jacp_madd(acc, x, yneg);                   // - P
  acc = jacp_madd(acc, fp_mul(x, beta), yneg);     // - phi(P)
  return acc.inf != 0;
}

struct xyzz_words { uint32_t w[4][12]; uint32_t inf; };
CG1_HD void xyzz_export(const xyzz& a, xyzz_words& o) {
  o.inf = a.inf;
  if (a.inf) {
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 12; ++j) o.w[c][j] = 0;
    return;
  }
  fp_to_host_words(a.X, o.w[0]); fp_to_host_words(a.Y, o.w[1]); fp_to_host_words(a.ZZ, o.w[2]); fp_to_host_words(a.ZZZ, o.w[3]);
}

}  // namespace cg1
