// BLS12-381 scalar field Fr (r = 0x73eda753...00000001, 255 bits): 4 x 64-bit limbs, Montgomery radix 2^256,
// header-only, ONE source for the host (csrc/shuffle_verify.cpp) and the device (csrc/kernels_rows.h: the same functions
// compiled __host__ __device__, so both sides produce the same bytes by construction; on the device only the product is tuned:
// 32-bit limbs, round 3 -- a verification needs a few thousand Fr products against ~10^5 Fp products).
// Backs the verifier-side scalar work of the shuffle argument:
// the reference does this with one Python `Scalar` object per operation over the Rust wheel
// (py_arkworks_bls12381-stubs/__init__.pyi:32-54; ipa.py:164-186,216,227-229; same_msm.py:155-182,213;
// grand_prod.py:139-170; same_perm.py:94-98), including one `inverse()` per vector element (util.py:51-54) --
// here a vector is inverted with one field inversion (Montgomery's trick).
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <vector>
#include "bls_consts.h"

#if defined(__HIPCC__)
#define CG1FR_HD __host__ __device__ inline
#else
#define CG1FR_HD static inline
#endif

namespace cg1fr {

typedef unsigned __int128 u128;

struct fr { uint64_t l[4]; };       // Montgomery form, canonical (< r)

CG1FR_HD fr fr_zero() { return fr{{0, 0, 0, 0}}; }
CG1FR_HD fr fr_one() { return fr{{cg1::H_FR_R1[0], cg1::H_FR_R1[1], cg1::H_FR_R1[2], cg1::H_FR_R1[3]}}; }
CG1FR_HD bool fr_is_zero(const fr& a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
CG1FR_HD bool fr_eq(const fr& a, const fr& b) {
  return ((a.l[0] ^ b.l[0]) | (a.l[1] ^ b.l[1]) | (a.l[2] ^ b.l[2]) | (a.l[3] ^ b.l[3])) == 0;
}

CG1FR_HD bool geq_r(const uint64_t a[4]) {
  for (int i = 3; i >= 0; --i) {
    if (a[i] != cg1::H_FR[i]) return a[i] > cg1::H_FR[i];
  }
  return true;
}
CG1FR_HD void sub_r(uint64_t a[4]) {
  u128 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)a[i] - cg1::H_FR[i] - borrow;
    a[i] = (uint64_t)d;
    borrow = (d >> 64) & 1;
  }
}

CG1FR_HD fr fr_add(const fr& a, const fr& b) {
  fr r;
  u128 c = 0;
  for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
  if (geq_r(r.l)) sub_r(r.l);           // a + b < 2r < 2^256: no carry out
  return r;
}
CG1FR_HD fr fr_neg(const fr& a) {
  if (fr_is_zero(a)) return a;
  fr r;
  u128 borrow = 0;
  for (int i = 0; i < 4; ++i) {
    u128 d = (u128)cg1::H_FR[i] - a.l[i] - borrow;
    r.l[i] = (uint64_t)d;
    borrow = (d >> 64) & 1;
  }
  return r;
}
CG1FR_HD fr fr_sub(const fr& a, const fr& b) { return fr_add(a, fr_neg(b)); }

// Montgomery product a * b / 2^256 mod r (CIOS)
CG1FR_HD fr fr_mul(const fr& a, const fr& b) {
#if defined(__HIP_DEVICE_COMPILE__)
  // device: the same CIOS over eight 32-bit limbs -- a step is one v_mad_u64_u32 and one 64-bit add (a 64 x 64 -> 128 product costs
  // four of them plus the carries of the 128-bit accumulations: 857 instructions per product against ~420 here).  The result is the
  // canonical representative either way: the same bytes as the 64-bit code below.
  uint32_t A[8], B[8], M[8], t[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    A[2 * i] = (uint32_t)a.l[i]; A[2 * i + 1] = (uint32_t)(a.l[i] >> 32);
    B[2 * i] = (uint32_t)b.l[i]; B[2 * i + 1] = (uint32_t)(b.l[i] >> 32);
    M[2 * i] = (uint32_t)cg1::H_FR[i]; M[2 * i + 1] = (uint32_t)(cg1::H_FR[i] >> 32);
  }
  const uint32_t ninv = (uint32_t)cg1::H_FR_INV;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const uint64_t p = (uint64_t)A[j] * B[i] + t[j] + c; t[j] = (uint32_t)p; c = (uint32_t)(p >> 32); }
    const uint64_t s = (uint64_t)t[8] + c;
    t[8] = (uint32_t)s;
    const uint32_t t9 = (uint32_t)(s >> 32);
    const uint32_t m = t[0] * ninv;
    c = (uint32_t)(((uint64_t)m * M[0] + t[0]) >> 32);
#pragma unroll
    for (int j = 1; j < 8; ++j) { const uint64_t p = (uint64_t)m * M[j] + t[j] + c; t[j - 1] = (uint32_t)p; c = (uint32_t)(p >> 32); }
    const uint64_t s2 = (uint64_t)t[8] + c;
    t[7] = (uint32_t)s2;
    t[8] = t9 + (uint32_t)(s2 >> 32);
  }
  fr r{{(uint64_t)t[0] | ((uint64_t)t[1] << 32), (uint64_t)t[2] | ((uint64_t)t[3] << 32), (uint64_t)t[4] | ((uint64_t)t[5] << 32),
        (uint64_t)t[6] | ((uint64_t)t[7] << 32)}};
  if (t[8] || geq_r(r.l)) sub_r(r.l);
  return r;
#else
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u128 c = 0;
    for (int j = 0; j < 4; ++j) {
      c += (u128)a.l[j] * b.l[i] + t[j];
      t[j] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[4] = (uint64_t)c;
    t[5] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * cg1::H_FR_INV;
    c = (u128)m * cg1::H_FR[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; ++j) {
      c += (u128)m * cg1::H_FR[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    c += t[4];
    t[3] = (uint64_t)c;
    t[4] = t[5] + (uint64_t)(c >> 64);
  }
  fr r{{t[0], t[1], t[2], t[3]}};
  if (t[4] || geq_r(r.l)) sub_r(r.l);
  return r;
#endif
}
CG1FR_HD fr fr_sqr(const fr& a) { return fr_mul(a, a); }

CG1FR_HD fr fr_from_u64(uint64_t v) {
  fr a{{v, 0, 0, 0}};
  fr r2{{cg1::H_FR_R2[0], cg1::H_FR_R2[1], cg1::H_FR_R2[2], cg1::H_FR_R2[3]}};
  return fr_mul(a, r2);
}
// 32 little-endian bytes, canonical (< r) required: false otherwise (Scalar.from_le_bytes raises ValueError,
// test_curdleproofs.py:210-213)
CG1FR_HD bool fr_from_le32(const uint8_t* b, fr& out) {
  fr a;
  memcpy(a.l, b, 32);
  if (geq_r(a.l)) return false;
  fr r2{{cg1::H_FR_R2[0], cg1::H_FR_R2[1], cg1::H_FR_R2[2], cg1::H_FR_R2[3]}};
  out = fr_mul(a, r2);
  return true;
}
// out of Montgomery form: four reduction rounds only (half the work of a multiplication by 1)
CG1FR_HD void fr_to_le32(const fr& a, uint8_t* b) {
  uint64_t t[4] = {a.l[0], a.l[1], a.l[2], a.l[3]};
  for (int i = 0; i < 4; ++i) {
    const uint64_t m = t[0] * cg1::H_FR_INV;
    u128 c = (u128)m * cg1::H_FR[0] + t[0];
    c >>= 64;
    for (int j = 1; j < 4; ++j) {
      c += (u128)m * cg1::H_FR[j] + t[j];
      t[j - 1] = (uint64_t)c;
      c >>= 64;
    }
    t[3] = (uint64_t)c;
  }
  if (geq_r(t)) sub_r(t);                 // t < r already for canonical input; kept for safety
  memcpy(b, t, 32);
}

CG1FR_HD fr fr_pow_u64(fr base, uint64_t e) {
  fr acc = fr_one();
  while (e) {
    if (e & 1) acc = fr_mul(acc, base);
    base = fr_sqr(base);
    e >>= 1;
  }
  return acc;
}

// a^(r-2); 0 -> 0
CG1FR_HD fr fr_inv(const fr& a) {
  fr acc = fr_one();
  for (int i = 254; i >= 0; --i) {
    acc = fr_sqr(acc);
    if ((cg1::H_FR_MINUS_2[i >> 6] >> (i & 63)) & 1) acc = fr_mul(acc, a);
  }
  return acc;
}

// in-place inversion of n NON-ZERO elements with one field inversion (host only)
static inline void fr_batch_inv(fr* v, size_t n) {
  if (n == 0) return;
  std::vector<fr> pre(n);
  fr acc = fr_one();
  for (size_t i = 0; i < n; ++i) { pre[i] = acc; acc = fr_mul(acc, v[i]); }
  fr inv = fr_inv(acc);
  for (size_t i = n; i-- > 0;) {
    fr t = fr_mul(inv, pre[i]);
    inv = fr_mul(inv, v[i]);
    v[i] = t;
  }
}

}  // namespace cg1fr
