// The endomorphism split of a scalar: k = k1 + k2 * lambda (mod r) with |k1|, |k2| < 2^127.
// Included inside namespace cg1 by csrc/msm_gpu.hip (device + host) and by the host library (lazy_host.cpp, for the CPU tests).
//
// BLS12-381 G1 has phi(x, y) = (beta x, y) with phi(P) = lambda P for every P of the PRIME-ORDER subgroup, lambda = z^2 - 1
// (lambda^2 + lambda + 1 = r; the same beta as the subgroup test [z^2]P = phi(P) + P of kernels_batch.h / lazy_host.cpp).  An MSM over
// n points certified in G1 is then an MSM over the 2n points P_i, phi(P_i) with 127-bit scalars: the same number of bucket additions,
// half the windows -- half the buckets to reduce and half the doublings of the Horner tail.  OUTSIDE the subgroup phi(P) != lambda P:
// the split is applied only where the caller vouches for the points (cg1_ctx_set_param "glv", the python face's certified leaves).
//
// The split (exact integers, no floating point; tools/gen_consts.py checks the constants):
//   k' = k - [k >= (r+1)/2] r         |k'| <= r/2          (k < 2^255 need not be canonical: k in [r, 2^255) gives k' < 0.11 r)
//   t = |k'| + floor(lambda / 2),  q = floor(t / lambda)   (one product with mu = floor(2^255 / lambda), off by at most one)
//   |k'| = q lambda + (t - q lambda - floor(lambda / 2))   =>   k2 = sign(k') q,  k1 = sign(k') (rem - floor(lambda / 2))
#pragma once

struct GlvParts {
  uint32_t k1[4], k2[4];               // magnitudes, < 2^127
  uint32_t neg1, neg2;
};

CG1_HD void glv_split(const uint32_t k[8], GlvParts& o) {
  constexpr uint32_t R[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
  constexpr uint32_t RH[8] = {0x80000001u, 0x7fffffffu, 0x7fff2dffu, 0xa9ded201u, 0x04d0ec02u, 0x199cec04u, 0x94cebea4u, 0x39f6d3a9u};   // (r + 1) / 2
  constexpr uint32_t LAM[4] = {0xffffffffu, 0x00000000u, 0x0001a402u, 0xac45a401u};
  constexpr uint32_t H[4] = {0x7fffffffu, 0x00000000u, 0x8000d201u, 0x5622d200u};                                                       // floor(lambda / 2)
  constexpr uint32_t MU[4] = {0x7b67f718u, 0xb1fb7291u, 0xf00fd56eu, 0xbe35f678u};                                                      // floor(2^255 / lambda)
  // k >= (r + 1) / 2 ?
  bool ge = true;
  for (int i = 7; i >= 0; --i) {
    if (k[i] != RH[i]) { ge = k[i] > RH[i]; break; }
  }
  uint32_t m[8];
  bool s = false;
  if (ge) {
    uint64_t br = 0;                                       // d = k - r
    for (int i = 0; i < 8; ++i) {
      const uint64_t v = (uint64_t)k[i] - R[i] - br;
      m[i] = (uint32_t)v; br = (v >> 32) & 1u;
    }
    if (br) {                                              // k < r: |k'| = r - k = -d
      s = true;
      uint64_t c = 1;
      for (int i = 0; i < 8; ++i) { const uint64_t v = (uint64_t)(~m[i]) + c; m[i] = (uint32_t)v; c = v >> 32; }
    }
  } else {
    for (int i = 0; i < 8; ++i) m[i] = k[i];
  }
  // t = m + H  (< 2^255)
  uint32_t t[8];
  {
    uint64_t c = 0;
    for (int i = 0; i < 8; ++i) { const uint64_t v = (uint64_t)m[i] + (i < 4 ? H[i] : 0u) + c; t[i] = (uint32_t)v; c = v >> 32; }
  }
  // prod = t * MU (12 words), q = prod >> 255
  uint32_t prod[12];
  for (int i = 0; i < 12; ++i) prod[i] = 0;
  for (int j = 0; j < 4; ++j) {
    uint64_t c = 0;
    for (int i = 0; i < 8; ++i) {
      const uint64_t v = (uint64_t)t[i] * MU[j] + prod[i + j] + c;
      prod[i + j] = (uint32_t)v; c = v >> 32;
    }
    prod[8 + j] = (uint32_t)c;
  }
  uint32_t q[4];
  for (int i = 0; i < 4; ++i) q[i] = (prod[7 + i] >> 31) | (prod[8 + i] << 1);
  // rem = t - q * LAM  (5 words are enough: 0 <= rem < 2 lambda < 2^129)
  uint32_t ql[5] = {0, 0, 0, 0, 0};
  for (int j = 0; j < 4; ++j) {
    uint64_t c = 0;
    for (int i = 0; i < 4 && i + j < 5; ++i) {
      const uint64_t v = (uint64_t)q[i] * LAM[j] + ql[i + j] + c;
      ql[i + j] = (uint32_t)v; c = v >> 32;
    }
    if (j == 0) ql[4] = (uint32_t)c;                       // (for j >= 1 the carry leaves the 160 bits kept)
  }
  uint32_t rem[5];
  {
    uint64_t br = 0;
    for (int i = 0; i < 5; ++i) { const uint64_t v = (uint64_t)t[i] - ql[i] - br; rem[i] = (uint32_t)v; br = (v >> 32) & 1u; }
  }
  // rem >= lambda: one correction
  bool big = rem[4] != 0;
  if (!big) {
    big = true;
    for (int i = 3; i >= 0; --i) {
      if (rem[i] != LAM[i]) { big = rem[i] > LAM[i]; break; }
    }
  }
  if (big) {
    uint64_t br = 0;
    for (int i = 0; i < 5; ++i) { const uint64_t v = (uint64_t)rem[i] - (i < 4 ? LAM[i] : 0u) - br; rem[i] = (uint32_t)v; br = (v >> 32) & 1u; }
    uint64_t c = 1;
    for (int i = 0; i < 4; ++i) { const uint64_t v = (uint64_t)q[i] + c; q[i] = (uint32_t)v; c = v >> 32; }
  }
  // k1 = rem - H (signed)
  bool lt = false;
  for (int i = 3; i >= 0; --i) {
    if (rem[i] != H[i]) { lt = rem[i] < H[i]; break; }
  }
  {
    uint64_t br = 0;
    for (int i = 0; i < 4; ++i) {
      const uint64_t v = lt ? (uint64_t)H[i] - rem[i] - br : (uint64_t)rem[i] - H[i] - br;
      o.k1[i] = (uint32_t)v; br = (v >> 32) & 1u;
    }
  }
  for (int i = 0; i < 4; ++i) o.k2[i] = q[i];
  o.neg1 = (s != lt) ? 1u : 0u;
  o.neg2 = s ? 1u : 0u;
}
