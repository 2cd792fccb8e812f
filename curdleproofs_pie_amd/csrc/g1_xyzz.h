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
