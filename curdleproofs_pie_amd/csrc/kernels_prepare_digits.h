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

// ------------------------------------------------------------------ signed digit recoding
// The 256 bit positions of a scalar are cut into nwin windows of widths width[w] <= cmax starting at bit off[w]
// (uniform: width = c, off = c w; balanced plans mix cmax and cmax - 1 so that EVERY window keeps >= cmax - 2 scalar bits:
// a thin top window would pile all n terms into a handful of buckets).  Digit w: value in [-(2^(width-1)-1), 2^(width-1)].
struct WinPlan {
  int nwin, cmax, n_hi;                  // windows 0 .. n_hi-1 have width cmax, the others cmax - 1 (uniform: n_hi = nwin)
  CG1_HD int width(int w) const { return w < n_hi ? cmax : cmax - 1; }
  CG1_HD int off(int w) const { return w < n_hi ? w * cmax : n_hi * cmax + (w - n_hi) * (cmax - 1); }
};
struct DigitIter {
  uint32_t s[8];
  uint32_t carry;
  __device__ __forceinline__ int next(const WinPlan& pl, int w) {         // must be called for w = 0,1,2,... in order
    const int bit = pl.off(w), c = pl.width(w);
    uint32_t wi = bit >> 5, sh = bit & 31;
    uint64_t v = (wi < 8) ? s[wi] : 0u;
    if (wi + 1 < 8) v |= (uint64_t)s[wi + 1] << 32;
    uint32_t raw = (uint32_t)(v >> sh) & ((1u << c) - 1u);
    uint32_t d = raw + carry;
    if (d > (1u << (c - 1))) { carry = 1; return (int)d - (1 << c); }
    carry = 0;
    return (int)d;
  }
};
__device__ __forceinline__ void load_scalar(const uint32_t* scalars, uint32_t i, DigitIter& it) {
  const uint4* q = reinterpret_cast<const uint4*>(scalars + 8ull * i);
  uint4 a = q[0], b = q[1];
  it.s[0] = a.x; it.s[1] = a.y; it.s[2] = a.z; it.s[3] = a.w;
  it.s[4] = b.x; it.s[5] = b.y; it.s[6] = b.z; it.s[7] = b.w;
  it.carry = 0;
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
  for (int w = 0; w < pl.nwin; ++w) {
    int d = it.next(pl, w);
    if (d == 0 || (w % world) != rank) continue;
    uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
    atomicAdd(&hist[(uint32_t)(w / world) * NB + b], 1u);
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
  for (int w = 0; w < pl.nwin; ++w) {
    int d = it.next(pl, w);
    if (d == 0 || (w % world) != rank) continue;
    uint32_t key = (uint32_t)(w / world) * NB + (uint32_t)(d < 0 ? -d : d) - 1u;
    uint32_t slot = atomicSub(&cursor[key], 1u) - 1u;      // cursor starts at the bucket's count
    sorted[off[key] + slot] = i | (d < 0 ? 0x80000000u : 0u);
  }
}

constexpr int SCAN_ITEMS = 1024;        // items per block of 256 threads in the scans

