// BLS12-381 base field Fp for the gfx950 kernels: 14 x 28-bit limbs held in u32, Montgomery radix 2^392.
//
// Why this shape (measured on MI355X, profiles/r01_ubench_valu_rates.txt):
//   * v_mad_u64_u32 (32x32+64 -> 64) sustains ~30 T lane-op/s chip-wide -- about the same cost as ANY
//     VOP3 op, while carry-flag adds (v_add_co/v_addc_co) are half rate, i.e. as dear as a multiply,
//     and plain v_add_u32 is full rate.  f64 FMA buys nothing (same rate, 3 ops per partial product).
//   * so: make every partial product one v_mad_u64_u32 into a 64-bit column accumulator and never touch
//     the carry flag.  28-bit limbs leave 8 spare bits per column: 14 a*b products + 14 q*p products
//     of <= 2^60 fit in 64 bits, so a whole Montgomery product needs no carry handling besides one
//     64-bit shift per column.  Add/sub are limb-wise full-rate adds with NO carry chain ("lazy"):
//     limbs may grow past 28 bits and values past p; bounds are tracked statically (see each routine)
//     and checked dynamically in the CG1_CHECK_BOUNDS host build (tests/native/).
//
// Conventions.  "N-form": limbs 0..12 < 2^28, limb 13 small, value < 2p -- what fp_mul returns.
// The code is plain C++ that compiles for the device (hipcc) and for the host (g++, bound-check tests).
#pragma once
#include <cstdint>
#include "bls_consts.h"

#if defined(__HIPCC__)
#define CG1_HD __host__ __device__ __forceinline__
#else
#define CG1_HD inline
#endif

#if defined(CG1_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
#include <cstdio>
#include <cstdlib>
#define CG1_ASSERT(c) do { if (!(c)) { fprintf(stderr, "CG1 bound violated: %s (%s:%d)\n", #c, __FILE__, __LINE__); abort(); } } while (0)
#else
#define CG1_ASSERT(c) ((void)0)
#endif

namespace cg1 {

constexpr int NL = 14;               // limbs
constexpr uint32_t LMASK = 0x0fffffffu;

struct fp { uint32_t l[NL]; };

// ---- constants as functions (constexpr arrays are not ODR-usable from device code without copies)
CG1_HD uint32_t c_p(int i)    { constexpr uint32_t t[NL] = {D_P[0], D_P[1], D_P[2], D_P[3], D_P[4], D_P[5], D_P[6], D_P[7], D_P[8], D_P[9], D_P[10], D_P[11], D_P[12], D_P[13]}; return t[i]; }

template <int K> struct kp_tab;      // padded multiples K*p (every limb >= 2^28-1 except the top), see gen_consts.py
#define CG1_KP_TAB(K) template <> struct kp_tab<K> { \
  static CG1_HD uint32_t get(int i) { constexpr uint32_t t[NL] = {D_KP##K[0], D_KP##K[1], D_KP##K[2], D_KP##K[3], D_KP##K[4], D_KP##K[5], D_KP##K[6], D_KP##K[7], D_KP##K[8], D_KP##K[9], D_KP##K[10], D_KP##K[11], D_KP##K[12], D_KP##K[13]}; return t[i]; } };
CG1_KP_TAB(3)
CG1_KP_TAB(6)
CG1_KP_TAB(12)
CG1_KP_TAB(32)
#undef CG1_KP_TAB

// acc + a*b.  With LLVM's Reassociate pass in the pipeline every column's MAD chain starts from zero and the shifted
// carry of the previous column is added with one extra v_lshl_add_u64 (~6 % of a Montgomery product); build.py leaves
// that pass out of the device pipeline, so the chain starts from the carry as written here.
CG1_HD uint64_t mad64(uint32_t a, uint32_t b, uint64_t c) {
#if defined(CG1_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
  unsigned __int128 w = (unsigned __int128)a * b + c;
  CG1_ASSERT((w >> 64) == 0);
#endif
  return (uint64_t)a * b + c;        // -> v_mad_u64_u32
}
CG1_HD uint64_t mad64c(uint32_t a, uint32_t b, uint64_t c) { return mad64(a, b, c); }   // b: a limb of p (an SGPR constant)

CG1_HD fp fp_zero() { fp r; for (int i = 0; i < NL; ++i) r.l[i] = 0; return r; }
CG1_HD fp fp_one()  {                // Montgomery form of 1 (N-form)
  constexpr uint32_t t[NL] = {D_R1[0], D_R1[1], D_R1[2], D_R1[3], D_R1[4], D_R1[5], D_R1[6], D_R1[7], D_R1[8], D_R1[9], D_R1[10], D_R1[11], D_R1[12], D_R1[13]};
  fp r; for (int i = 0; i < NL; ++i) r.l[i] = t[i]; return r;
}
CG1_HD fp fp_r2()   {
  constexpr uint32_t t[NL] = {D_R2[0], D_R2[1], D_R2[2], D_R2[3], D_R2[4], D_R2[5], D_R2[6], D_R2[7], D_R2[8], D_R2[9], D_R2[10], D_R2[11], D_R2[12], D_R2[13]};
  fp r; for (int i = 0; i < NL; ++i) r.l[i] = t[i]; return r;
}

// Montgomery product, product scanning.  Requires max_limb(a) * max_limb(b) <= 2^60 (e.g. both < 2^30)
// and value(a)*value(b) < p * 2^392 (e.g. both < 48p).  Returns N-form, value < p + a*b/2^392.
CG1_HD fp fp_mul(const fp& a, const fp& b) {
  fp r;
  uint32_t q[NL];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) acc = mad64(a.l[i], b.l[k - i], acc);
#pragma unroll
    for (int i = 0; i < k; ++i) acc = mad64c(q[i], c_p(k - i), acc);
    q[k] = ((uint32_t)acc * D_PINV) & LMASK;
    acc = mad64c(q[k], c_p(0), acc);
    acc >>= 28;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc = mad64(a.l[i], b.l[k - i], acc);
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc = mad64c(q[i], c_p(k - i), acc);
    r.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= 28;
  }
  CG1_ASSERT((acc >> 32) == 0);
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

// Fused (a*b + c*d) * 2^-392: both products share ONE Montgomery reduction (saves 196 MADs + the second
// normalisation).  Requires 14*(max_limb(a)*max_limb(b) + max_limb(c)*max_limb(d)) + 14*2^56 < 2^64
// (checked by mad64 in the bound-check build) and value(a)*value(b) + value(c)*value(d) < p * 2^392.
CG1_HD fp fp_mul2(const fp& a, const fp& b, const fp& c, const fp& d) {
  fp r;
  uint32_t q[NL];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) { acc = mad64(a.l[i], b.l[k - i], acc); acc = mad64(c.l[i], d.l[k - i], acc); }
#pragma unroll
    for (int i = 0; i < k; ++i) acc = mad64c(q[i], c_p(k - i), acc);
    q[k] = ((uint32_t)acc * D_PINV) & LMASK;
    acc = mad64c(q[k], c_p(0), acc);
    acc >>= 28;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) { acc = mad64(a.l[i], b.l[k - i], acc); acc = mad64(c.l[i], d.l[k - i], acc); }
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc = mad64c(q[i], c_p(k - i), acc);
    r.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= 28;
  }
  CG1_ASSERT((acc >> 32) == 0);
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

// Montgomery square: the symmetric a_i*a_j terms are taken once against a doubled operand.
// Requires max_limb(a) < 2^30 (doubled limb < 2^31; 7 cross terms + 1 square + 14 q*p per column).
CG1_HD fp fp_sqr(const fp& a) {
  fp r;
  uint32_t q[NL], a2[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) a2[i] = a.l[i] << 1;
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < NL; ++k) {
#pragma unroll
    for (int i = 0; 2 * i < k; ++i) acc = mad64(a.l[i], a2[k - i], acc);
    if ((k & 1) == 0) acc = mad64(a.l[k / 2], a.l[k / 2], acc);
#pragma unroll
    for (int i = 0; i < k; ++i) acc = mad64c(q[i], c_p(k - i), acc);
    q[k] = ((uint32_t)acc * D_PINV) & LMASK;
    acc = mad64c(q[k], c_p(0), acc);
    acc >>= 28;
  }
#pragma unroll
  for (int k = NL; k < 2 * NL - 1; ++k) {
#pragma unroll
    for (int i = k - NL + 1; 2 * i < k; ++i) acc = mad64(a.l[i], a2[k - i], acc);
    if ((k & 1) == 0) acc = mad64(a.l[k / 2], a.l[k / 2], acc);
#pragma unroll
    for (int i = k - NL + 1; i < NL; ++i) acc = mad64c(q[i], c_p(k - i), acc);
    r.l[k - NL] = (uint32_t)acc & LMASK;
    acc >>= 28;
  }
  CG1_ASSERT((acc >> 32) == 0);
  r.l[NL - 1] = (uint32_t)acc;
  return r;
}

// Lazy add: limb-wise, no carries.  max_limb grows additively; caller keeps it under the fp_mul bound.
CG1_HD fp fp_add(const fp& a, const fp& b) {
  fp r;
#pragma unroll
  for (int i = 0; i < NL; ++i) { r.l[i] = a.l[i] + b.l[i]; CG1_ASSERT(r.l[i] >= a.l[i]); }
  return r;
}
CG1_HD fp fp_dbl(const fp& a) { return fp_add(a, a); }

// Lazy subtract a - b + K*p.  Requires b in N-limb form (limbs 0..12 < 2^28) with value < (K-1)*p.
// Result limbs < max_limb(a) + 2^29, value < value(a) + K*p.
template <int K>
CG1_HD fp fp_sub(const fp& a, const fp& b) {
  fp r;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    CG1_ASSERT(kp_tab<K>::get(i) >= b.l[i]);
    r.l[i] = a.l[i] + (kp_tab<K>::get(i) - b.l[i]);
    CG1_ASSERT(r.l[i] >= a.l[i]);
  }
  return r;
}
// K*p - b  (negation), same requirement on b.
template <int K>
CG1_HD fp fp_neg(const fp& b) {
  fp r;
#pragma unroll
  for (int i = 0; i < NL; ++i) { CG1_ASSERT(kp_tab<K>::get(i) >= b.l[i]); r.l[i] = kp_tab<K>::get(i) - b.l[i]; }
  return r;
}

// Carry-propagate so that limbs 0..12 < 2^28 (value unchanged; limb 13 keeps the excess).
CG1_HD fp fp_norm(const fp& a) {
  fp r;
  uint32_t c = 0;
#pragma unroll
  for (int i = 0; i < NL - 1; ++i) {
    uint32_t t = a.l[i] + c;
    CG1_ASSERT(t >= c);
    r.l[i] = t & LMASK;
    c = t >> 28;
  }
  r.l[NL - 1] = a.l[NL - 1] + c;
  CG1_ASSERT(r.l[NL - 1] >= c);
  return r;
}

// Exact "value == 0 mod p" for a lazily reduced a with value < KMAX*p, KMAX <= 64.
// Fast path: a = k*p  =>  (a_0 * p^-1) mod 2^28 = k < KMAX; anything else is certainly non-zero.
CG1_HD bool fp_is_zero_mod_p(const fp& a, uint32_t kmax) {
  // D_PINV = -p^-1 mod 2^28, so k = (-(a_0 * D_PINV)) mod 2^28
  uint32_t k = (0u - a.l[0] * D_PINV) & LMASK;
  if (k >= kmax) return false;
  // slow path (probability ~kmax/2^28 for random data): compare the normalised limbs with k*p
  fp n = fp_norm(a);
  uint64_t c = 0;
  bool eq = true;
  for (int i = 0; i < NL; ++i) {
    c += (uint64_t)k * c_p(i);
    uint32_t want = (i < NL - 1) ? ((uint32_t)c & LMASK) : (uint32_t)c;
    if (i < NL - 1) c >>= 28;
    eq = eq && (n.l[i] == want);
  }
  return eq;
}

// raw 48-byte little-endian integer (12 u32 words, value < p) -> limbs (standard, not Montgomery)
CG1_HD fp fp_from_words(const uint32_t w[12]) {
  fp r;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int bit = 28 * i, wi = bit >> 5, sh = bit & 31;
    uint64_t v = w[wi];
    if (wi + 1 < 12) v |= (uint64_t)w[wi + 1] << 32;
    r.l[i] = (uint32_t)(v >> sh) & LMASK;
  }
  return r;
}
CG1_HD fp fp_to_mont(const fp& std_form) { return fp_mul(std_form, fp_r2()); }

// Montgomery -> canonical standard integer in [0,p), packed into 12 u32 words.
// Requires value(a) < 2^392 (always true under the fp_mul input bounds).
CG1_HD void fp_to_words(const fp& a, uint32_t w[12]) {
  fp one = fp_zero(); one.l[0] = 1;
  fp t = fp_mul(a, one);             // value in [0, p]
  bool is_p = true;
#pragma unroll
  for (int i = 0; i < NL; ++i) is_p = is_p && (t.l[i] == c_p(i));
  if (is_p) t = fp_zero();
#pragma unroll
  for (int j = 0; j < 12; ++j) w[j] = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int bit = 28 * i, wi = bit >> 5, sh = bit & 31;
    uint64_t v = (uint64_t)t.l[i] << sh;
    w[wi] |= (uint32_t)v;
    if (wi + 1 < 12) w[wi + 1] |= (uint32_t)(v >> 32);
  }
}

// Montgomery (radix 2^392) -> the HOST library's Montgomery form (radix 2^384), canonical, packed into 12 u32 words:
// montmul(a * 2^392, 2^384 mod p) = a * 2^384.  Saves the host one multiplication per exported coordinate.
CG1_HD void fp_to_host_words(const fp& a, uint32_t w[12]) {
  constexpr uint32_t kt[NL] = {D_HOSTR[0], D_HOSTR[1], D_HOSTR[2], D_HOSTR[3], D_HOSTR[4], D_HOSTR[5], D_HOSTR[6], D_HOSTR[7], D_HOSTR[8], D_HOSTR[9], D_HOSTR[10], D_HOSTR[11], D_HOSTR[12], D_HOSTR[13]};
  fp k; for (int i = 0; i < NL; ++i) k.l[i] = kt[i];
  fp t = fp_mul(a, k);               // N-form, value < p + a*k/2^392 < 2p
  bool ge = true;                    // t >= p ?
  for (int i = NL - 1; i >= 0; --i) {
    if (t.l[i] != c_p(i)) { ge = t.l[i] > c_p(i); break; }
  }
  if (ge) {
    uint32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      uint32_t d = t.l[i] - c_p(i) - borrow;
      borrow = (i < NL - 1) ? (d >> 31) : 0u;        // limbs < 2^28: a negative difference sets bit 31
      t.l[i] = (i < NL - 1) ? (d & LMASK) : d;
    }
  }
#pragma unroll
  for (int j = 0; j < 12; ++j) w[j] = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int bit = 28 * i, wi = bit >> 5, sh = bit & 31;
    uint64_t v = (uint64_t)t.l[i] << sh;
    w[wi] |= (uint32_t)v;
    if (wi + 1 < 12) w[wi + 1] |= (uint32_t)(v >> 32);
  }
}

// a^e for a 384-bit exponent given as 6 x 64-bit words (square-and-multiply, MSB first).  Off the hot path.
// Sliding window of width 3 over the (wave-uniform, constant) exponent: odd powers a, a^3, a^5, a^7 in registers,
// ~380 squarings + ~96 multiplications for a 381-bit exponent instead of ~190 with square-and-multiply.
CG1_HD fp fp_pow6(const fp& a, const uint64_t e[6]) {
  const fp a2 = fp_sqr(a);
  const fp t3 = fp_mul(a, a2), t5 = fp_mul(t3, a2), t7 = fp_mul(t5, a2);
  auto bit = [&](int i) -> unsigned { return (unsigned)((e[i >> 6] >> (i & 63)) & 1u); };
  fp r = fp_one();
  bool started = false;
  int i = 383;
  while (i >= 0 && !bit(i)) --i;
  while (i >= 0) {
    if (!bit(i)) { r = fp_sqr(r); --i; continue; }
    int j = i >= 2 ? i - 2 : 0;
    while (!bit(j)) ++j;                                   // window [i .. j], odd value
    unsigned val = 0;
    for (int t = i; t >= j; --t) val = (val << 1) | bit(t);
    fp m;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const uint32_t lo = (val & 2u) ? t3.l[l] : a.l[l], hi = (val & 2u) ? t7.l[l] : t5.l[l];
      m.l[l] = (val & 4u) ? hi : lo;
    }
    if (started) {
      for (int t = i; t >= j; --t) r = fp_sqr(r);
      r = fp_mul(r, m);
    } else {
      r = m;
      started = true;
    }
    i = j - 1;
  }
  return r;
}
// a^(p-2): inversion by Fermat (affine outputs of generated points).
CG1_HD fp fp_inv(const fp& a) {
  constexpr uint64_t e[6] = {H_INV_EXP[0], H_INV_EXP[1], H_INV_EXP[2], H_INV_EXP[3], H_INV_EXP[4], H_INV_EXP[5]};
  return fp_pow6(a, e);
}
// a^((p+1)/4): the square root candidate (p = 3 mod 4); the caller checks r^2 == a.
CG1_HD fp fp_sqrt_candidate(const fp& a) {
  constexpr uint64_t e[6] = {H_SQRT_EXP[0], H_SQRT_EXP[1], H_SQRT_EXP[2], H_SQRT_EXP[3], H_SQRT_EXP[4], H_SQRT_EXP[5]};
  return fp_pow6(a, e);
}

}  // namespace cg1
