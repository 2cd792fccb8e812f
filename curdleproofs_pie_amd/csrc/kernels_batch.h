// Batched scalar-mul / fold, batched G1 decompression, synthetic scalars, madd probe.
// Part of the single translation unit csrc/msm_gpu.hip (included inside namespace cg1).
#pragma once

// ------------------------------------------------------------------ batched scalar mul / fold
// out[i] = addend[i] + k_i * P_i   with P_i = base[i % nbase], k_i = scalars[i % nscalars], addend optional.
// Covers the vectorised `G1Point * Scalar` patterns of the callers:
//   nbase = 1                  fixed base:        get_random_point = G * random_scalar()   (util.py:67-68)
//   nscalars = 1               same-scalar map:   [R * k for R in vec_R]                   (curdleproofs.py:310-311)
//   nscalars = 1, addend = L   fold:              G_L[i] + G_R[i] * gamma                  (ipa.py:142-146, same_msm.py:122-126)
//   per-index scalars          G_i * beta^-i                                               (grand_prod.py:64-71)
// Affine std words in and out (identity = zeros); one lane per output, double-and-add MSB first.
__global__ void __launch_bounds__(128) k_batch_mul(const uint32_t* __restrict__ base_raw, uint32_t nbase,
                                                   const uint32_t* __restrict__ scalars, uint32_t nscalars,
                                                   const uint32_t* __restrict__ addend_raw, uint32_t* __restrict__ out_raw, uint32_t n) {
  uint32_t i = blockIdx.x * 128 + threadIdx.x;
  if (i >= n) return;
  uint32_t w[24];
  const uint32_t* src = base_raw + 24ull * (i % nbase);
  uint32_t any = 0;
  for (int k = 0; k < 24; ++k) { w[k] = src[k]; any |= w[k]; }
  uint32_t s[8];
  for (int k = 0; k < 8; ++k) s[k] = scalars[8ull * (i % nscalars) + k];
  xyzz acc = xyzz_identity();
  if (any) {
    fp x = fp_to_mont(fp_from_words(w)), y = fp_to_mont(fp_from_words(w + 12));
    int top = 255;
    while (top >= 0 && !((s[top >> 5] >> (top & 31)) & 1u)) --top;     // skip leading zero bits (per lane)
    for (int bit = top; bit >= 0; --bit) {
      acc = xyzz_dbl(acc);
      if ((s[bit >> 5] >> (bit & 31)) & 1u) acc = xyzz_madd(acc, x, y);
    }
  }
  if (addend_raw) {
    const uint32_t* a = addend_raw + 24ull * i;
    uint32_t aw[24], aany = 0;
    for (int k = 0; k < 24; ++k) { aw[k] = a[k]; aany |= aw[k]; }
    if (aany) acc = xyzz_madd(acc, fp_to_mont(fp_from_words(aw)), fp_to_mont(fp_from_words(aw + 12)));
  }
  uint32_t* dst = out_raw + 24ull * i;
  if (acc.inf) { for (int k = 0; k < 24; ++k) dst[k] = 0; return; }
  // x = X/ZZ, y = Y/ZZZ with ONE inversion: 1/(ZZ*ZZZ)
  fp t = fp_inv(fp_mul(acc.ZZ, acc.ZZZ));
  fp izz = fp_mul(t, acc.ZZZ), izzz = fp_mul(t, acc.ZZ);
  uint32_t o[12];
  fp_to_words(fp_mul(acc.X, izz), o);  for (int k = 0; k < 12; ++k) dst[k] = o[k];
  fp_to_words(fp_mul(acc.Y, izzz), o); for (int k = 0; k < 12; ++k) dst[12 + k] = o[k];
}

// The same map for SMALL n (the prover's fold launches: <= a few hundred outputs each, prover_kernels.py): the launch is a pure
// latency chain, so one DPP QUAD per output (g1_quad.h: doubling 3.5 instead of 8 multiply-times, addition 4.5 instead of 14.5)
// and two scalar bits per step over the table {P, 2P, 3P}: 128 x (2 doublings + 1 addition) instead of 255 x (doubling + mixed
// addition whenever any lane of the wave has the bit set).  Same outputs, bit for bit (tests/test_prover_kernels_gpu.py).
__global__ void __launch_bounds__(64) k_batch_mul_quad(const uint32_t* __restrict__ base_raw, uint32_t nbase,
                                                       const uint32_t* __restrict__ scalars, uint32_t nscalars,
                                                       const uint32_t* __restrict__ addend_raw, uint32_t* __restrict__ out_raw, uint32_t n) {
  const uint32_t t = blockIdx.x * 64 + threadIdx.x, i = t >> 2, q = t & 3u;
  if (i >= n) return;                                   // whole quads leave together
  uint32_t w[24];
  const uint32_t* src = base_raw + 24ull * (i % nbase);
  uint32_t any = 0;
  for (int k = 0; k < 24; ++k) { w[k] = src[k]; any |= w[k]; }
  uint32_t s[8];
  for (int k = 0; k < 8; ++k) s[k] = scalars[8ull * (i % nscalars) + k];
  xyzz acc = xyzz_identity();
  if (any) {
    const xyzz P1 = xyzz_from_affine(fp_to_mont(fp_from_words(w)), fp_to_mont(fp_from_words(w + 12)));
    const xyzz P2 = quad_dbl(P1, q);
    const xyzz P3 = quad_add(P2, P1, q);
#pragma unroll 1
    for (int pair = 127; pair >= 0; --pair) {
      acc = quad_dbl(quad_dbl(acc, q), q);
      const uint32_t d = (s[pair >> 4] >> ((pair & 15) * 2)) & 3u;
      if (d) {                                            // uniform inside the quad (its 4 lanes hold the same scalar)
        xyzz T;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
          T.X.l[k] = d == 1u ? P1.X.l[k] : (d == 2u ? P2.X.l[k] : P3.X.l[k]);
          T.Y.l[k] = d == 1u ? P1.Y.l[k] : (d == 2u ? P2.Y.l[k] : P3.Y.l[k]);
          T.ZZ.l[k] = d == 1u ? P1.ZZ.l[k] : (d == 2u ? P2.ZZ.l[k] : P3.ZZ.l[k]);
          T.ZZZ.l[k] = d == 1u ? P1.ZZZ.l[k] : (d == 2u ? P2.ZZZ.l[k] : P3.ZZZ.l[k]);
        }
        T.inf = 0;
        acc = quad_add(acc, T, q);
      }
    }
  }
  if (addend_raw) {
    const uint32_t* a = addend_raw + 24ull * i;
    uint32_t aw[24], aany = 0;
    for (int k = 0; k < 24; ++k) { aw[k] = a[k]; aany |= aw[k]; }
    if (aany) acc = quad_add(acc, xyzz_from_affine(fp_to_mont(fp_from_words(aw)), fp_to_mont(fp_from_words(aw + 12))), q);
  }
  if (q) return;
  uint32_t* dst = out_raw + 24ull * i;
  if (acc.inf) { for (int k = 0; k < 24; ++k) dst[k] = 0; return; }
  fp tinv = fp_inv(fp_mul(acc.ZZ, acc.ZZZ));
  fp izz = fp_mul(tinv, acc.ZZZ), izzz = fp_mul(tinv, acc.ZZ);
  uint32_t o[12];
  fp_to_words(fp_mul(acc.X, izz), o);  for (int k = 0; k < 12; ++k) dst[k] = o[k];
  fp_to_words(fp_mul(acc.Y, izzz), o); for (int k = 0; k < 12; ++k) dst[12 + k] = o[k];
}

// ------------------------------------------------------------------ segmented point sum (fourth pattern)
// G_sum = reduce(lambda a, b: a + b, vec_G, Z1), H_sum likewise (crs.py:64-65): group j is the sum of the points
// [offs[j], offs[j+1]) of the input.  One wave per group: the lanes stride over the group's points with mixed additions
// (every exceptional case exact: equal points, opposite points, identities), a shuffle tree joins the 64 lane sums, lane 0
// normalises with one inversion.  Affine std words in and out (identity = zeros).
__global__ void __launch_bounds__(64) k_batch_sum(const uint32_t* __restrict__ base_raw, const uint32_t* __restrict__ offs,
                                                  uint32_t* __restrict__ out_raw) {
  const uint32_t g = blockIdx.x, lo = offs[g], hi = offs[g + 1];
  xyzz acc = xyzz_identity();
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 64) {
    const uint32_t* src = base_raw + 24ull * i;
    uint32_t w[24], any = 0;
    for (int k = 0; k < 24; ++k) { w[k] = src[k]; any |= w[k]; }
    if (any) acc = xyzz_madd(acc, fp_to_mont(fp_from_words(w)), fp_to_mont(fp_from_words(w + 12)));
  }
  for (int delta = 32; delta >= 1; delta >>= 1) {
    xyzz o = shfl_down_xyzz(acc, delta);
    if (threadIdx.x < (uint32_t)delta) acc = xyzz_add(acc, o);
  }
  if (threadIdx.x) return;
  uint32_t* dst = out_raw + 24ull * g;
  if (acc.inf) { for (int k = 0; k < 24; ++k) dst[k] = 0; return; }
  fp t = fp_inv(fp_mul(acc.ZZ, acc.ZZZ));
  fp izz = fp_mul(t, acc.ZZZ), izzz = fp_mul(t, acc.ZZ);
  uint32_t o[12];
  fp_to_words(fp_mul(acc.X, izz), o);  for (int k = 0; k < 12; ++k) dst[k] = o[k];
  fp_to_words(fp_mul(acc.Y, izzz), o); for (int k = 0; k < 12; ++k) dst[12 + k] = o[k];
}

// ------------------------------------------------------------------ v_mad_u64_u32 issue-rate probe (bench.py, same-run peak)
// Eight independent 64-bit accumulator chains per lane, nothing but the multiply-add the field arithmetic is made of: the
// chip-wide rate this sustains at 2 waves per SIMD is the `peak` of roofline_int_mad, measured on the box and at the clock
// the benchmark itself runs at (tools/ubench_valu.hip is the stand-alone form with the other instructions beside it).
__global__ void __launch_bounds__(256) k_probe_mad_rate(uint32_t* __restrict__ out, int iters, uint32_t seed) {
  uint32_t a = seed * (threadIdx.x + 1) | 1u, b = seed ^ (0x9e3779b9u * (blockIdx.x + 1));
  uint64_t acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = (uint64_t)a * (c + 3) + b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < 8; ++c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
    }
  }
  uint64_t r = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) r ^= acc[c];
  if (r == 0x123456789abcdefull) out[0] = (uint32_t)r;      // keeps the chains alive; practically never taken
}

// ------------------------------------------------------------------ batched 48-byte G1 decompression
// One lane per point: parse the ZCash-format encoding (util.py:35-36 -> G1Point.from_compressed_bytes[_unchecked]),
// y = sqrt(x^3 + 4) by exponentiation, sign select, optional subgroup test  [z^2]P == phi(P) + P.
// out: affine96 (zeros = identity), status: 0 ok, CG1_ERR_ENCODING / _NOT_ON_CURVE / _NOT_IN_SUBGROUP.
// ------------------------------------------------------------------ batched 48-byte G1 compression
// affine96 (x || y little-endian standard form, zeros = identity) -> ZCash-format encoding (util.py:27-28 ->
// G1Point.to_compressed_bytes): big-endian x, bit 7 compressed, bit 6 infinity, bit 5 "y > (p-1)/2".  No field
// arithmetic: the inputs are already affine (k_batch_mul / k_batch_decompress produce this form).
__global__ void __launch_bounds__(256) k_batch_compress(const uint32_t* __restrict__ in_raw, uint8_t* __restrict__ out48, uint32_t n) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t* src = in_raw + 24ull * i;
  uint32_t w[24], any = 0;
  for (int k = 0; k < 24; ++k) { w[k] = src[k]; any |= w[k]; }
  uint8_t* o = out48 + 48ull * i;
  if (!any) { o[0] = 0xC0; for (int k = 1; k < 48; ++k) o[k] = 0; return; }
  bool is_large = false, decided = false;       // y > (p-1)/2 ?
  for (int j = 11; j >= 0 && !decided; --j) if (w[12 + j] != W_P_MINUS_1_HALF[j]) { is_large = w[12 + j] > W_P_MINUS_1_HALF[j]; decided = true; }
  for (int j = 0; j < 12; ++j) {                // little-endian words -> big-endian bytes
    uint32_t v = w[11 - j];
    o[4 * j] = (uint8_t)(v >> 24); o[4 * j + 1] = (uint8_t)(v >> 16); o[4 * j + 2] = (uint8_t)(v >> 8); o[4 * j + 3] = (uint8_t)v;
  }
  o[0] |= (uint8_t)(0x80 | (is_large ? 0x20 : 0));
}

// (g1_in_subgroup: g1_xyzz.h -- [z^2] P == phi(P) + P on Jacobian doublings)

// Subgroup flags for SELECTED points of each proof of a batch (affine96 standard-form words as k_batch_decompress wrote
// them; `stride_pts` points per proof; the k = so.k points at offsets so.off[] of every proof): flags[proof * k + j] = 1 if
// the point is NOT in G1.  The reference decodes unchecked (util.py:35-36) but asserts the same-scalar equalities EXACTLY
// (same_scalar.py:108); batching them under random weights is only sound for points of G1, so the verifier needs to know.
// Only ~10 points per proof: a latency-bound launch (a few hundred waves), so each point is tested by a DPP QUAD
// (g1_quad.h: 4 lanes share the field multiplications of every EC operation, ~2.6x shorter dependent chain) and the
// launch runs on the context's side stream, beside the kernels of the compute stream.
struct SgOffsets { uint32_t off[16]; uint32_t k; };
__global__ void __launch_bounds__(64) k_subgroup_flags(const uint32_t* __restrict__ aff, uint32_t stride_pts, uint32_t n_proofs,
                                                       SgOffsets so, uint8_t* __restrict__ flags) {
  const uint32_t t = (blockIdx.x * 64 + threadIdx.x) >> 2, q = threadIdx.x & 3u;
  const uint32_t proof = t / so.k, j = t % so.k;
  if (proof >= n_proofs) return;                    // whole quads leave together
  const uint32_t* src = aff + 24ull * ((size_t)proof * stride_pts + so.off[j]);
  uint32_t w[24], any = 0;
  for (int k = 0; k < 24; ++k) { w[k] = src[k]; any |= w[k]; }
  if (!any) { if (q == 0) flags[t] = 0; return; }   // identity (or a point the decoder rejected: its proof is rejected anyway)
  const fp x = fp_to_mont(fp_from_words(w)), y = fp_to_mont(fp_from_words(w + 12));
  constexpr uint64_t ZABS = 0xd201000000010000ull;
  constexpr uint32_t bt[NL] = {D_BETA[0], D_BETA[1], D_BETA[2], D_BETA[3], D_BETA[4], D_BETA[5], D_BETA[6], D_BETA[7], D_BETA[8], D_BETA[9], D_BETA[10], D_BETA[11], D_BETA[12], D_BETA[13]};
  fp beta; for (int k = 0; k < NL; ++k) beta.l[k] = bt[k];
  // acc = [|z|] [|z|] P - P - phi(P) as ONE loop with one quad_dbl and one quad_add call site (instruction cache, registers):
  // steps 0..62 and 63..125: double, then add the pass's base (P, then Q = [|z|] P) where |z| has a one; steps 126, 127: add -P, -phi(P).
  const fp yneg = fp_norm(fp_neg<3>(y));
  const fp bx = fp_mul(x, beta);
  xyzz a = xyzz_from_affine(x, y), base = a;
#pragma unroll 1
  for (int step = 0; step < 128; ++step) {
    bool add = true;
    if (step < 126) {
      if (step == 63) base = a;                    // second pass: the base is Q
      const int bit = 62 - (step < 63 ? step : step - 63);
      a = quad_dbl(a, q);
      add = (ZABS >> bit) & 1ull;
    } else {
      base = xyzz_from_affine(step == 126 ? x : bx, yneg);
    }
    if (add) a = quad_add(a, base, q);
  }
  if (q == 0) flags[t] = a.inf ? 0 : 1;
}

// y = a^((p+1)/4) by the FIXED addition chain of tools/sqrt_chain.py (bls_consts.h D_SQRT_CHAIN): sliding window of width 4
// over the constant exponent, 376 squarings + 85 products (the run-time 3-bit window of fp_pow6 needed 378 + 109 and a
// per-limb select chain for every product).  The eight odd powers a, a^3, ..., a^15 stay in registers; the chain steps come
// from constant memory through scalar loads, so the step's table index is wave-uniform and picking the operand is a
// scalar branch over plain register copies: the loop body has ONE squaring and ONE product call site (small code, no
// divergence, no select chains).  (An LDS table -- 448 B per lane -- would halve the occupancy.)
constexpr int SQRT_TAB = 8;
__constant__ uint16_t g_sqrt_chain[D_SQRT_CHAIN_LEN] = CG1_SQRT_CHAIN_INIT;
__device__ __forceinline__ fp sqrt_tab_pick(const fp (&T)[SQRT_TAB], uint32_t idx) {       // idx is wave-uniform
  switch (idx) {
    case 0: return T[0]; case 1: return T[1]; case 2: return T[2]; case 3: return T[3];
    case 4: return T[4]; case 5: return T[5]; case 6: return T[6]; default: return T[7];
  }
}
__device__ __forceinline__ fp fp_sqrt_chain(const fp& a) {
  const fp a2 = fp_sqr(a);
  fp T[SQRT_TAB];
  T[0] = a;
  fp cur = a;
#pragma unroll 1
  for (int i = 1; i < SQRT_TAB; ++i) {
    cur = fp_mul(cur, a2);
    switch (i) {                                     // uniform: a scalar branch around register copies
      case 1: T[1] = cur; break; case 2: T[2] = cur; break; case 3: T[3] = cur; break; case 4: T[4] = cur; break;
      case 5: T[5] = cur; break; case 6: T[6] = cur; break; default: T[7] = cur; break;
    }
  }
  fp r = T[D_SQRT_CHAIN_FIRST];
#pragma unroll 1
  for (int k = 0; k < D_SQRT_CHAIN_LEN; ++k) {
    const uint32_t op = g_sqrt_chain[k];
#pragma unroll 1
    for (uint32_t n = op >> 8; n; --n) r = fp_sqr(r);
    const uint32_t idx = op & 0xffu;
    if (idx != 0xffu) r = fp_mul(r, sqrt_tab_pick(T, idx));
  }
  return r;
}

// CHECK is a template parameter: the unchecked instantiation (the reference's default, util.py:35-36) must not carry the
// register footprint of the subgroup test's scalar multiplication (256 VGPRs + spills, 1 wave/SIMD when it did).
// WAVES (per SIMD) is a template parameter too: at 3 the compiler keeps half of the chain's table in scratch (168 VGPRs, 224 bytes, 45
// scratch loads per point against 148 862 multiplies) and a third wave shares the multiplier; 2 = the table in registers (214 VGPRs).
template <bool CHECK, int WAVES>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) k_batch_decompress(const uint8_t* __restrict__ in48, uint32_t* __restrict__ out_raw,
                                                          uint8_t* __restrict__ status, uint32_t n) {
  uint32_t i = blockIdx.x * 128 + threadIdx.x;
  if (i >= n) return;
  const uint8_t* b = in48 + 48ull * i;
  uint32_t* dst = out_raw + 24ull * i;
  for (int k = 0; k < 24; ++k) dst[k] = 0;
  const uint8_t flags = b[0];
  const bool compressed = flags & 0x80, infinity = flags & 0x40, largest = flags & 0x20;
  uint32_t w[12];
  for (int j = 0; j < 12; ++j) {                // big-endian bytes -> little-endian words
    const uint8_t* q = b + 44 - 4 * j;
    uint32_t b0 = (j == 11) ? (uint32_t)(q[0] & 0x1F) : (uint32_t)q[0];
    w[j] = (b0 << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
  }
  if (!compressed) { status[i] = CG1_ERR_ENCODING; return; }
  if (infinity) { status[i] = CG1_OK; return; }          // the identity whatever the other bits say (as the wheel decodes it; host_g1.cpp)
  bool lt = false, decided = false;             // x < p ?
  for (int j = 11; j >= 0 && !decided; --j) if (w[j] != W_P[j]) { lt = w[j] < W_P[j]; decided = true; }
  if (!lt) { status[i] = CG1_ERR_ENCODING; return; }
  const fp x = fp_to_mont(fp_from_words(w));
  fp four = fp_one(); four = fp_dbl(fp_dbl(four));
  const fp rhs = fp_norm(fp_add(fp_mul(fp_sqr(x), x), four));
  fp y = fp_sqrt_chain(rhs);
  if (!fp_is_zero_mod_p(fp_sub<3>(fp_sqr(y), fp_mul(rhs, fp_one())), 8)) { status[i] = CG1_ERR_NOT_ON_CURVE; return; }
  uint32_t yw[12];
  fp_to_words(y, yw);
  bool is_large = false; decided = false;       // y > (p-1)/2 ?
  for (int j = 11; j >= 0 && !decided; --j) if (yw[j] != W_P_MINUS_1_HALF[j]) { is_large = yw[j] > W_P_MINUS_1_HALF[j]; decided = true; }
  if (is_large != largest) {                     // y := p - y  (y != 0: the curve has no point with y = 0)
    uint64_t borrow = 0;
    for (int j = 0; j < 12; ++j) {
      uint64_t d = (uint64_t)W_P[j] - yw[j] - borrow;
      yw[j] = (uint32_t)d; borrow = (d >> 32) & 1;
    }
    y = fp_to_mont(fp_from_words(yw));
  }
  if (CHECK) {
    if (!g1_in_subgroup(x, y)) { status[i] = CG1_ERR_NOT_IN_SUBGROUP; return; }
  }
  for (int k = 0; k < 12; ++k) { dst[k] = w[k]; dst[12 + k] = yw[k]; }
  status[i] = CG1_OK;
}

// The same decoding for a few hundred points at a time (the 585 single from_compressed_bytes_unchecked of one verification, decoded in
// one batch by the deferred G1Point layer): ONE DPP ROW per point, four points per wave, the square-root chain with one limb per lane
// (fp_row.h: a product in ~0.2 us where the one-lane form needs ~0.9, so the chain of 461 takes ~0.1 ms instead of ~0.4; the host's
// worker pool needs 0.58 ms for 585 points).  Parsing, x^3 + 4, the check y^2 = x^3 + 4 and the sign choice run in the one-lane form,
// redundantly in the 16 lanes of the row; lane 0 of the row writes.  No subgroup test (the unchecked decoding).
__global__ void __launch_bounds__(64) k_batch_decompress_row(const uint8_t* __restrict__ in48, uint32_t* __restrict__ out_raw,
                                                             uint8_t* __restrict__ status, uint32_t n) {
  const uint32_t lane = threadIdx.x & 63u, row = lane >> 4;
  const uint32_t i = blockIdx.x * 4u + row;
  const RowK k = row_constants();
  const bool have = i < n;
  uint32_t w[12];
  uint32_t st = CG1_OK;
  bool live = false, largest = false;
  fp rhs = fp_one();                                     // rows without a point to decode run the chain on 1
  if (have) {
    const uint8_t* b = in48 + 48ull * i;
    const uint8_t flags = b[0];
    const bool compressed = flags & 0x80, infinity = flags & 0x40;
    largest = flags & 0x20;
    for (int j = 0; j < 12; ++j) {                // big-endian bytes -> little-endian words
      const uint8_t* q = b + 44 - 4 * j;
      const uint32_t b0 = (j == 11) ? (uint32_t)(q[0] & 0x1F) : (uint32_t)q[0];
      w[j] = (b0 << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
    }
    if (!compressed) st = CG1_ERR_ENCODING;
    else if (!infinity) {
      bool lt = false, decided = false;             // x < p ?
      for (int j = 11; j >= 0 && !decided; --j) if (w[j] != W_P[j]) { lt = w[j] < W_P[j]; decided = true; }
      if (!lt) st = CG1_ERR_ENCODING;
      else {
        live = true;
        const fp x = fp_to_mont(fp_from_words(w));
        fp four = fp_one(); four = fp_dbl(fp_dbl(four));
        rhs = fp_norm(fp_add(fp_mul(fp_sqr(x), x), four));
      }
    }
  }
  // ---- y = rhs^((p+1)/4): the chain of fp_sqrt_chain, on rows (every lane of the wave takes part)
  const uint32_t a = row_from_fp(rhs, k.lane16);
  const uint32_t a2 = row_mul(a, a, k);
  uint32_t T0 = a, T1, T2, T3, T4, T5, T6, T7;
  T1 = row_mul(T0, a2, k); T2 = row_mul(T1, a2, k); T3 = row_mul(T2, a2, k); T4 = row_mul(T3, a2, k);
  T5 = row_mul(T4, a2, k); T6 = row_mul(T5, a2, k); T7 = row_mul(T6, a2, k);
  static_assert(D_SQRT_CHAIN_FIRST == 6, "the chain starts from a^13");
  uint32_t r = T6;
#pragma unroll 1
  for (int s = 0; s < D_SQRT_CHAIN_LEN; ++s) {
    const uint32_t op = g_sqrt_chain[s];
#pragma unroll 1
    for (uint32_t m = op >> 8; m; --m) r = row_mul(r, r, k);
    const uint32_t idx = op & 0xffu;                     // wave-uniform
    if (idx != 0xffu) {
      uint32_t t;
      switch (idx) {
        case 0: t = T0; break; case 1: t = T1; break; case 2: t = T2; break; case 3: t = T3; break;
        case 4: t = T4; break; case 5: t = T5; break; case 6: t = T6; break; default: t = T7; break;
      }
      r = row_mul(r, t, k);
    }
  }
  fp y = fp_norm(row_to_fp(r));
  // ---- the one-lane tail
  if (!have) return;
  uint32_t* dst = out_raw + 24ull * i;
  uint32_t yw[12];
  for (int j = 0; j < 12; ++j) yw[j] = 0;
  if (live) {
    if (!fp_is_zero_mod_p(fp_sub<3>(fp_sqr(y), fp_mul(rhs, fp_one())), 8)) { st = CG1_ERR_NOT_ON_CURVE; live = false; }
    else {
      fp_to_words(y, yw);
      bool is_large = false, decided = false;       // y > (p-1)/2 ?
      for (int j = 11; j >= 0 && !decided; --j) if (yw[j] != W_P_MINUS_1_HALF[j]) { is_large = yw[j] > W_P_MINUS_1_HALF[j]; decided = true; }
      if (is_large != largest) {                     // y := p - y
        uint64_t borrow = 0;
        for (int j = 0; j < 12; ++j) {
          const uint64_t d = (uint64_t)W_P[j] - yw[j] - borrow;
          yw[j] = (uint32_t)d; borrow = (d >> 32) & 1;
        }
      }
    }
  }
  if (k.lane16 == 0u) {
    for (int j = 0; j < 12; ++j) { dst[j] = live ? w[j] : 0u; dst[12 + j] = live ? yw[j] : 0u; }
    status[i] = (uint8_t)st;
  }
}

// splitmix64-derived scalars, uniform in [1, r-1] (the reference's random_scalar distribution,
// util.py:21-24) by rejection from 255-bit draws; deterministic in (seed, i)
__global__ void __launch_bounds__(256) k_gen_scalars(uint32_t* __restrict__ out, uint32_t n, uint64_t seed) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t st = seed + 0x9E3779B97F4A7C15ull * (64ull * i + 1);
  uint64_t v[4];
  for (int attempt = 0; attempt < 64; ++attempt) {
    for (int k = 0; k < 4; ++k) {
      st += 0x9E3779B97F4A7C15ull;
      uint64_t z = st;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      v[k] = z ^ (z >> 31);
    }
    v[3] &= 0x7FFFFFFFFFFFFFFFull;       // 255 bits
    bool lt = false, decided = false;     // v < r ?
    for (int k = 3; k >= 0 && !decided; --k) {
      if (v[k] != H_FR[k]) { lt = v[k] < H_FR[k]; decided = true; }
    }
    bool nz = (v[0] | v[1] | v[2] | v[3]) != 0;
    if (lt && nz) break;
    if (attempt == 63) { v[3] = 0; v[0] |= 1; }   // unreachable in practice (p ~ 2^-64)
  }
  for (int k = 0; k < 4; ++k) { out[8ull * i + 2 * k] = (uint32_t)v[k]; out[8ull * i + 2 * k + 1] = (uint32_t)(v[k] >> 32); }
}

// throughput probe: `iters` dependent mixed adds per lane on register-resident data (roofline of k_accumulate)
__global__ void __launch_bounds__(256) k_probe_madd(const PreparedPoint* __restrict__ pts, uint32_t npts, PointSum* __restrict__ out, int iters) {
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  fp x, y; uint32_t flags;
  load_affine(pts + (t % npts), x, y, flags);
  fp x2, y2;
  load_affine(pts + ((t + 1) % npts), x2, y2, flags);
  xyzz acc = xyzz_from_affine(x, y);
  for (int i = 0; i < iters; ++i) acc = xyzz_madd(acc, x2, y2);
  store_sum(out + t, acc);
}

