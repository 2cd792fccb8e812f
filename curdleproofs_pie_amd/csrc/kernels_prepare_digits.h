// k_prepare_points, signed-digit recoding, the global-atomic counting sort (k_hist / k_scatter; n > 2^23) and k_digits.
// Part of the single translation unit csrc/msm_gpu.hip (included inside namespace cg1).
#pragma once

// ------------------------------------------------------------------ k_prepare_points
// 96-B affine record -> 128-B Montgomery record.  A block moves its 256 records through LDS so that both the reads
// (24.6 KB) and the writes (32 KB) are whole contiguous lines; a lane-strided access touched 64 lines per instruction
// (3.5 TB/s effective; this kernel runs over ALL N * 2^20 points on every rank of a window-sharded MSM).
// (also clears the call's status words -- the input-validation flag k_digits may set -- so no memset launch is needed)
__global__ void __launch_bounds__(256) k_prepare_points(const uint32_t* __restrict__ raw, PreparedPoint* __restrict__ out,
                                                        uint8_t* __restrict__ inf_flag, uint32_t n, uint32_t* __restrict__ status_words = nullptr) {
  if (status_words && blockIdx.x == 0 && threadIdx.x < 4) status_words[threadIdx.x] = 0;
  __shared__ uint4 stage[256 * 8];                       // 32 KB: input (6 uint4 per record), then output (8 per record)
  const uint32_t base = blockIdx.x * 256u, t = threadIdx.x;
  const uint32_t cnt = (n - base < 256u) ? n - base : 256u;
  const uint4* src = reinterpret_cast<const uint4*>(raw + 24ull * base);
  for (uint32_t j = t; j < cnt * 6u; j += 256u) stage[j] = src[j];
  __syncthreads();
  uint32_t o[32];
  if (t < cnt) {
    uint32_t w[24];
#pragma unroll
    for (int k = 0; k < 6; ++k) { uint4 v = stage[t * 6u + k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < 24; ++k) any |= w[k];
    fp x = fp_to_mont(fp_from_words(w));
    fp y = fp_to_mont(fp_from_words(w + 12));
#pragma unroll
    for (int k = 0; k < NL; ++k) { o[k] = x.l[k]; o[NL + k] = y.l[k]; }
    o[28] = any ? 0u : 1u;               // (0,0) is not on the curve: it encodes the identity
    inf_flag[base + t] = any ? 0 : 1;    // compact copy: the digit kernels must not touch the 128-B records
    o[29] = o[30] = o[31] = 0;
  }
  __syncthreads();                       // everyone has consumed its input record
  if (t < cnt) {
#pragma unroll
    for (int k = 0; k < 8; ++k) stage[t * 8u + k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
  }
  __syncthreads();
  uint4* dst = reinterpret_cast<uint4*>(out + base);
  for (uint32_t j = t; j < cnt * 8u; j += 256u) dst[j] = stage[j];
}

// ------------------------------------------------------------------ k_prepare_blobs
// The same 128-B records straight from the HOST library's point blobs (144 B: Jacobian X | Y | Z, each 6 x 64-bit limbs in
// Montgomery radix 2^384 -- what a G1Point object holds, csrc/host_g1.h), so that compute_MSM(bases, scalars) over lists of
// G1Point (msm_accumulator.py:6-12) needs no normalisation pass on the host: the blobs are uploaded as they are.
//   NORM = false: every blob has Z = 1 or Z = 0 (decoded / generated points): two radix conversions per point.
//   NORM = true:  x = X / Z^2, y = Y / Z^3 with Montgomery's trick over the K consecutive points of a lane: ONE inversion
//                 (Fermat, ~470 products) per lane + 11 products per point.  The running products and the converted Z of
//                 the forward pass are parked in the point's own output record (read back by the same lane in the backward
//                 pass before the record gets its final contents).
// A coordinate word-triple v = X * 2^384 mod p (an integer < p) becomes the device's Montgomery form with one product:
// montmul(v, 2^400 mod p) = v * 2^8 = X * 2^392.
__device__ __forceinline__ fp fp_from_host_words(const uint32_t w[12]) {
  constexpr uint32_t kt[NL] = {D_H2D[0], D_H2D[1], D_H2D[2], D_H2D[3], D_H2D[4], D_H2D[5], D_H2D[6], D_H2D[7], D_H2D[8], D_H2D[9], D_H2D[10], D_H2D[11], D_H2D[12], D_H2D[13]};
  fp k; for (int i = 0; i < NL; ++i) k.l[i] = kt[i];
  return fp_mul(fp_from_words(w), k);
}
__device__ __forceinline__ void load_words12(const uint32_t* p, uint32_t w[12]) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int k = 0; k < 3; ++k) { uint4 v = q[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
}
__device__ __forceinline__ void store_fp14(uint32_t* dst, const fp& a) {      // 14 words at a 8-byte aligned address
  uint2* q = reinterpret_cast<uint2*>(dst);
#pragma unroll
  for (int k = 0; k < 7; ++k) q[k] = make_uint2(a.l[2 * k], a.l[2 * k + 1]);
}
__device__ __forceinline__ fp load_fp14(const uint32_t* src) {
  const uint2* q = reinterpret_cast<const uint2*>(src);
  fp a;
#pragma unroll
  for (int k = 0; k < 7; ++k) { uint2 v = q[k]; a.l[2 * k] = v.x; a.l[2 * k + 1] = v.y; }
  return a;
}
template <bool NORM>
__global__ void __launch_bounds__(128) k_prepare_blobs(const uint32_t* __restrict__ blobs, PreparedPoint* __restrict__ out,
                                                       uint8_t* __restrict__ inf_flag, uint32_t n, uint32_t K,
                                                       uint32_t* __restrict__ status_words) {
  if (status_words && blockIdx.x == 0 && threadIdx.x < 4) status_words[threadIdx.x] = 0;
  const uint32_t t = blockIdx.x * 128 + threadIdx.x;
  const uint64_t first = (uint64_t)t * K;
  if (first >= n) return;
  const uint32_t i0 = (uint32_t)first, i1 = (n - i0 < K) ? n : i0 + K;
  uint32_t w[12];
  if (!NORM) {
    for (uint32_t i = i0; i < i1; ++i) {
      const uint32_t* b = blobs + 36ull * i;
      uint32_t* o = reinterpret_cast<uint32_t*>(out + i);
      load_words12(b + 24, w);
      uint32_t zany = 0;
#pragma unroll
      for (int k = 0; k < 12; ++k) zany |= w[k];
      load_words12(b, w);
      const fp x = fp_from_host_words(w);
      load_words12(b + 12, w);
      const fp y = fp_from_host_words(w);
      store_fp14(o, zany ? x : fp_zero());
      store_fp14(o + NL, zany ? y : fp_zero());
      reinterpret_cast<uint4*>(o)[7] = make_uint4(zany ? 0u : 1u, 0u, 0u, 0u);
      inf_flag[i] = zany ? 0 : 1;
    }
    return;
  }
  // forward: running product of the non-zero Z's; record i keeps the product BEFORE its own Z (x slot) and its converted Z (y slot)
  fp acc = fp_one();
  for (uint32_t i = i0; i < i1; ++i) {
    uint32_t* o = reinterpret_cast<uint32_t*>(out + i);
    load_words12(blobs + 36ull * i + 24, w);
    uint32_t zany = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) zany |= w[k];
    inf_flag[i] = zany ? 0 : 1;
    if (!zany) continue;
    const fp z = fp_from_host_words(w);
    store_fp14(o, acc);
    store_fp14(o + NL, z);
    acc = fp_mul(acc, z);
  }
  fp inv = fp_inv(acc);                                   // (acc = 1 when every Z of the lane is 0: harmless)
  for (uint32_t i = i1; i-- > i0;) {
    uint32_t* o = reinterpret_cast<uint32_t*>(out + i);
    if (inf_flag[i]) {
      store_fp14(o, fp_zero()); store_fp14(o + NL, fp_zero());
      reinterpret_cast<uint4*>(o)[7] = make_uint4(1u, 0u, 0u, 0u);
      continue;
    }
    const fp pref = load_fp14(o), z = load_fp14(o + NL);
    const fp zi = fp_mul(inv, pref);                      // 1 / Z_i
    inv = fp_mul(inv, z);
    const fp zi2 = fp_sqr(zi);
    load_words12(blobs + 36ull * i, w);
    const fp x = fp_mul(fp_from_host_words(w), zi2);
    load_words12(blobs + 36ull * i + 12, w);
    const fp y = fp_mul(fp_from_host_words(w), fp_mul(zi2, zi));
    store_fp14(o, x);
    store_fp14(o + NL, y);
    reinterpret_cast<uint4*>(o)[7] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// ------------------------------------------------------------------ signed digit recoding
// The 256 bit positions of a scalar are cut into nwin windows of widths width[w] <= cmax starting at bit off[w]
// (uniform: width = c, off = c w; balanced plans mix cmax and cmax - 1 so that EVERY window keeps >= cmax - 2 scalar bits:
// a thin top window would pile all n terms into a handful of buckets).  Digit w: value in [-(2^(width-1)-1), 2^(width-1)].
struct WinPlan {
  int nwin, cmax, n_hi;                  // windows 0 .. n_hi-1 have width cmax, the others cmax - 1 (uniform: n_hi = nwin)
  int glv;                               // 1: the windows cut the 127-bit halves of the endomorphism split (glv.h); the digit kernel then
                                         // writes TWO rows of digits per scalar, for the points i and n + i (= phi(P_i)) of a 2n-record table
  CG1_HD int width(int w) const { return w < n_hi ? cmax : cmax - 1; }
  CG1_HD int off(int w) const { return w < n_hi ? w * cmax : n_hi * cmax + (w - n_hi) * (cmax - 1); }
};
// Which windows of the plan one launch chain takes, as the (rank, sel) pair the kernels carry:
//   sel = world | lw0 << 8 | cnt << 16
//   the share: windows w = rank (mod world) -- the per-GPU shares of a window-sharded MSM; world = 1: every window;
//   of the share's windows (numbered 0, 1, ... in ascending w) the chain takes the cnt consecutive ones from lw0 (cnt = 0: all of
//   them) -- the two halves of ONE call that run as two chains on two streams.
// win_local: the chain's local index of global window w, or -1 when the window is not its own.
CG1_HD int win_local(int w, int rank, int sel) {
  const int world = sel & 0xff, lw0 = (sel >> 8) & 0xff, cnt = (sel >> 16) & 0xff;
  if ((w % world) != rank) return -1;
  const int lw = w / world - lw0;
  return (lw >= 0 && (cnt == 0 || lw < cnt)) ? lw : -1;
}
CG1_HD int win_global(int lw, int rank, int sel) { return rank + (lw + ((sel >> 8) & 0xff)) * (sel & 0xff); }
CG1_HD int win_count(int nwin, int rank, int sel) {
  const int world = sel & 0xff, lw0 = (sel >> 8) & 0xff, cnt = (sel >> 16) & 0xff;
  return cnt ? cnt : (nwin - rank + world - 1) / world - lw0;
}
CG1_HD int win_sel(int world, int lw0, int cnt) { return world | (lw0 << 8) | (cnt << 16); }
// The same test for w = 0, 1, 2, ... in order, without a division per window (the digit kernels walk all windows of every scalar:
// win_local's `w % world` cost k_digits 14 of its 52 us at 2^20 terms).  step(w) returns the chain's local index of w, or -1.
struct WinWalk {
  int own, lw, world, cnt;
  CG1_HD WinWalk(int rank, int sel) : own(rank), lw(-((sel >> 8) & 0xff)), world(sel & 0xff), cnt((sel >> 16) & 0xff) {}
  CG1_HD int step(int w) {
    if (w != own) return -1;
    const int l = lw;
    own += world; ++lw;
    return (l >= 0 && (cnt == 0 || l < cnt)) ? l : -1;
  }
};

struct DigitIter {
  uint32_t s[8];
  uint32_t carry;
  uint32_t flip;                         // 1: the digits of the NEGATED value: the tie d = 2^(c-1) goes the other way, so that the negated
                                         // digit stays in the u16 range (-2^15 has no encoding; 0x8000 means +2^15)
  // must be called for w = 0, 1, 2, ... in order: the window's bits are the low c bits of s, which is then shifted down by c as a
  // whole (eight v_alignbit with a wave-uniform count).  Picking the words by the window's bit offset instead indexes a register array
  // with a run-time index: k_digits took 31 us for 2^16 scalars of a 20-window plan that way, 3x what the shifts need.
  __device__ __forceinline__ int next(const WinPlan& pl, int w) {
    const int c = pl.width(w);                                             // 1 <= c <= 16
    const uint32_t raw = s[0] & ((1u << c) - 1u);
#pragma unroll
    for (int k = 0; k < 7; ++k) s[k] = __builtin_amdgcn_alignbit(s[k + 1], s[k], (uint32_t)c);
    s[7] >>= c;
    uint32_t d = raw + carry;
    int r;
    if (d > (1u << (c - 1)) - flip) { carry = 1; r = (int)d - (1 << c); }
    else { carry = 0; r = (int)d; }
    return flip ? -r : r;
  }
};
__device__ __forceinline__ void load_scalar(const uint32_t* scalars, uint32_t i, DigitIter& it) {
  const uint4* q = reinterpret_cast<const uint4*>(scalars + 8ull * i);
  uint4 a = q[0], b = q[1];
  it.s[0] = a.x; it.s[1] = a.y; it.s[2] = a.z; it.s[3] = a.w;
  it.s[4] = b.x; it.s[5] = b.y; it.s[6] = b.z; it.s[7] = b.w;
  it.carry = 0; it.flip = 0;
}
// one 127-bit half of the endomorphism split as a digit source
__device__ __forceinline__ void load_half(const uint32_t (&mag)[4], uint32_t neg, DigitIter& it) {
  it.s[0] = mag[0]; it.s[1] = mag[1]; it.s[2] = mag[2]; it.s[3] = mag[3];
  it.s[4] = it.s[5] = it.s[6] = it.s[7] = 0;
  it.carry = 0; it.flip = neg;
}

// ------------------------------------------------------------------ k_phi_records
// record n + i = phi(record i) = (beta x, y): the second half of the 2n-record table of an endomorphism-split MSM (glv.h).
__global__ void __launch_bounds__(256) k_phi_records(PreparedPoint* __restrict__ pts, uint8_t* __restrict__ inf_flag, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  constexpr uint32_t bt[NL] = {D_BETA[0], D_BETA[1], D_BETA[2], D_BETA[3], D_BETA[4], D_BETA[5], D_BETA[6], D_BETA[7], D_BETA[8], D_BETA[9], D_BETA[10], D_BETA[11], D_BETA[12], D_BETA[13]};
  fp beta; for (int k = 0; k < NL; ++k) beta.l[k] = bt[k];
  const uint4* src = reinterpret_cast<const uint4*>(pts + i);
  uint32_t w[32];
#pragma unroll
  for (int k = 0; k < 8; ++k) { const uint4 v = src[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
  fp x; for (int k = 0; k < NL; ++k) x.l[k] = w[k];
  const fp bx = fp_norm(fp_mul(x, beta));
  for (int k = 0; k < NL; ++k) w[k] = bx.l[k];
  uint4* dst = reinterpret_cast<uint4*>(pts + n + i);
#pragma unroll
  for (int k = 0; k < 8; ++k) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
  inf_flag[n + i] = inf_flag[i];
}

// counts per (local window, bucket); skips zero digits and identity points
__global__ void __launch_bounds__(256) k_hist(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf_flag,
                                              uint32_t* __restrict__ hist, uint32_t n, WinPlan pl, int rank, int world,
                                              uint32_t* __restrict__ bad_flag) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (inf_flag[i]) return;
  DigitIter it;
  load_scalar(scalars, i, it);
  if (it.s[7] >> 31) *bad_flag = 1u;
  const uint32_t NB = 1u << (pl.cmax - 1);
  WinWalk ww(rank, world);
  for (int w = 0; w < pl.nwin; ++w) {
    int d = it.next(pl, w);
    const int lw = ww.step(w);
    if (d == 0 || lw < 0) continue;
    uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
    atomicAdd(&hist[(uint32_t)lw * NB + b], 1u);
  }
}

__global__ void __launch_bounds__(256) k_scatter(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf_flag,
                                                 uint32_t* __restrict__ cursor, const uint32_t* __restrict__ off,
                                                 uint32_t* __restrict__ sorted, uint32_t n, WinPlan pl, int rank, int world) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (inf_flag[i]) return;
  DigitIter it;
  load_scalar(scalars, i, it);
  const uint32_t NB = 1u << (pl.cmax - 1);
  WinWalk ww(rank, world);
  for (int w = 0; w < pl.nwin; ++w) {
    int d = it.next(pl, w);
    const int lw = ww.step(w);
    if (d == 0 || lw < 0) continue;
    uint32_t key = (uint32_t)lw * NB + (uint32_t)(d < 0 ? -d : d) - 1u;
    uint32_t slot = atomicSub(&cursor[key], 1u) - 1u;      // cursor starts at the bucket's count
    sorted[off[key] + slot] = i | (d < 0 ? 0x80000000u : 0u);
  }
}

constexpr int SCAN_ITEMS = 1024;        // items per block of 256 threads in the scans

