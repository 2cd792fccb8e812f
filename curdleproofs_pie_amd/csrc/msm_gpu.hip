// MI355X (gfx950) Pippenger MSM over BLS12-381 G1 -- kernels and the per-device pipeline context.
//
// Replaces the hot loop of the reference's compute_MSM
//   (/root/reference/curdleproofs/curdleproofs/msm_accumulator.py:6-12:  current += base * scalar)
// with a signed-digit windowed-bucket method laid out for CDNA4:
//
//   k_prepare_points   96 B affine (std form)  -> 128 B records (x,y as 14x28-bit Montgomery limbs): one
//                      aligned cache line per point, so the bucket gather below touches exactly one line.
//   k_hist             per scalar: c-bit signed digits for this rank's windows -> per-bucket counts
//   k_scan1/2/3        exclusive scan of (count, chunk count) per bucket  (chunk = <= Lb consecutive
//                      entries of one bucket, Lb = max(L0, count/4096) so a skewed bucket -- e.g. the
//                      reference's [beta]*ell all-equal scalars, same_perm.py:54-55 -- is split into many
//                      pieces whose sums k_heavy_combine adds with a block-wide tree)
//   k_scatter          counting-sort scatter: sorted[] = point index | sign<<31, grouped by bucket
//   k_chunk_desc       (start,len) per chunk
//   k_accumulate  ***  the dominant kernel: one lane per chunk, XYZZ mixed adds (8M+2S) over gathered points
//   k_seg_reduce       per 4-bucket segment: run = sum B_k, tot = sum t*B_k   (running sums, depth 8)
//   k_bit_tree         per (window, bit b of the segment index): tree-sum of the selected seg runs, in
//                      registers -> wave shuffles -> LDS; emits canonical XYZZ words
//   host tail          <= 256 doublings of Horner over <= nwin*(c-2) points (host_g1.cpp): a single lane's
//                      EC op latency is ~20 us on the GPU vs ~0.5 us on a host core, so the strictly
//                      serial tail belongs on the host.
//
// No MFMA: this is carry-propagating big-integer arithmetic.  The bound is the VALU integer-multiply
// rate (v_mad_u64_u32), see fp28.h; HBM traffic is reported against the 8 TB/s roofline by bench.py.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include <chrono>
#include "g1_xyzz.h"
#include "host_g1.h"
#include "../../include/curdle_g1.h"

namespace cg1 {

// ------------------------------------------------------------------ device records
struct alignas(16) PreparedPoint {      // 128 B
  uint32_t x[NL];
  uint32_t y[NL];
  uint32_t flags;                       // bit0: identity
  uint32_t pad[3];
};
static_assert(sizeof(PreparedPoint) == 128, "one cache line per point");

struct alignas(16) PointSum {           // 256 B: an XYZZ partial sum
  uint32_t c[4][NL];
  uint32_t inf;
  uint32_t pad[7];
};
static_assert(sizeof(PointSum) == 256, "");

struct alignas(16) PointWords {         // 208 B: canonical standard-form XYZZ (see xyzz_words)
  uint32_t w[4][12];
  uint32_t inf;
  uint32_t pad[3];
};
static_assert(sizeof(PointWords) == 208, "");

__device__ __forceinline__ void load_affine(const PreparedPoint* p, fp& x, fp& y, uint32_t& flags) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = q[i];
  const uint32_t* w = reinterpret_cast<const uint32_t*>(v);
#pragma unroll
  for (int i = 0; i < NL; ++i) { x.l[i] = w[i]; y.l[i] = w[NL + i]; }
  flags = w[2 * NL];
}

__device__ __forceinline__ void store_sum(PointSum* dst, const xyzz& a) {
  uint32_t w[64];
#pragma unroll
  for (int i = 0; i < NL; ++i) { w[i] = a.X.l[i]; w[NL + i] = a.Y.l[i]; w[2 * NL + i] = a.ZZ.l[i]; w[3 * NL + i] = a.ZZZ.l[i]; }
  w[4 * NL] = a.inf;
#pragma unroll
  for (int i = 4 * NL + 1; i < 64; ++i) w[i] = 0;
  uint4* q = reinterpret_cast<uint4*>(dst);
#pragma unroll
  for (int i = 0; i < 16; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

__device__ __forceinline__ xyzz load_sum(const PointSum* src) {
  const uint4* q = reinterpret_cast<const uint4*>(src);
  uint32_t w[60];
#pragma unroll
  for (int i = 0; i < 15; ++i) { uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
  xyzz a;
#pragma unroll
  for (int i = 0; i < NL; ++i) { a.X.l[i] = w[i]; a.Y.l[i] = w[NL + i]; a.ZZ.l[i] = w[2 * NL + i]; a.ZZZ.l[i] = w[3 * NL + i]; }
  a.inf = w[4 * NL];
  return a;
}

// ------------------------------------------------------------------ k_prepare_points
__global__ void __launch_bounds__(256) k_prepare_points(const uint32_t* __restrict__ raw, PreparedPoint* __restrict__ out,
                                                        uint8_t* __restrict__ inf_flag, uint32_t n) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint4* q = reinterpret_cast<const uint4*>(raw + 24ull * i);
  uint32_t w[24];
#pragma unroll
  for (int k = 0; k < 6; ++k) { uint4 v = q[k]; w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w; }
  uint32_t any = 0;
#pragma unroll
  for (int k = 0; k < 24; ++k) any |= w[k];
  fp x = fp_to_mont(fp_from_words(w));
  fp y = fp_to_mont(fp_from_words(w + 12));
  uint32_t o[32];
#pragma unroll
  for (int k = 0; k < NL; ++k) { o[k] = x.l[k]; o[NL + k] = y.l[k]; }
  o[28] = any ? 0u : 1u;               // (0,0) is not on the curve: it encodes the identity
  inf_flag[i] = any ? 0 : 1;           // compact copy: the digit kernels must not touch the 128-B records
  o[29] = o[30] = o[31] = 0;
  uint4* d = reinterpret_cast<uint4*>(out + i);
#pragma unroll
  for (int k = 0; k < 8; ++k) d[k] = make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]);
}

// ------------------------------------------------------------------ signed digit recoding
// digit w of scalar s (LE words), window width c: value in [-(2^(c-1)-1), 2^(c-1)]
struct DigitIter {
  uint32_t s[8];
  uint32_t carry;
  int c;
  __device__ __forceinline__ int next(int w) {         // must be called for w = 0,1,2,... in order
    int bit = w * c;
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
                                              uint32_t* __restrict__ hist, uint32_t n, int c, int nwin, int rank, int world) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (inf_flag[i]) return;
  DigitIter it; it.c = c;
  load_scalar(scalars, i, it);
  const uint32_t NB = 1u << (c - 1);
  for (int w = 0; w < nwin; ++w) {
    int d = it.next(w);
    if (d == 0 || (w % world) != rank) continue;
    uint32_t b = (uint32_t)(d < 0 ? -d : d) - 1u;
    atomicAdd(&hist[(uint32_t)(w / world) * NB + b], 1u);
  }
}

__global__ void __launch_bounds__(256) k_scatter(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf_flag,
                                                 uint32_t* __restrict__ cursor, const uint32_t* __restrict__ off,
                                                 uint32_t* __restrict__ sorted, uint32_t n, int c, int nwin, int rank, int world) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (inf_flag[i]) return;
  DigitIter it; it.c = c;
  load_scalar(scalars, i, it);
  const uint32_t NB = 1u << (c - 1);
  for (int w = 0; w < nwin; ++w) {
    int d = it.next(w);
    if (d == 0 || (w % world) != rank) continue;
    uint32_t key = (uint32_t)(w / world) * NB + (uint32_t)(d < 0 ? -d : d) - 1u;
    uint32_t slot = atomicSub(&cursor[key], 1u) - 1u;      // cursor starts at the bucket's count
    sorted[off[key] + slot] = i | (d < 0 ? 0x80000000u : 0u);
  }
}

constexpr int SCAN_ITEMS = 1024;        // items per block of 256 threads in the scans

// ------------------------------------------------------------------ two-level partition sort (no global atomics)
// Replaces k_hist/k_scatter (16.7 M global atomics each at n = 2^20: 0.63 + 0.87 ms) when n <= 2^23.
//   k_digits        scalar -> one u16 signed digit per owned window, window-major  [nlw][n]
//   k_part_count    block = (window, tile of PART_TILE points): LDS histogram over the HIGH bits of the bucket
//                   ("bin"), written as block_counts[window][bin][tile]
//   k_uscan1/2/3    exclusive scan of block_counts: lexicographic (window, bin, tile) order = final layout
//   k_part_scatter  same blocks: LDS cursors seeded from the scan, entries (sub | sign | idx) land bin-grouped
//   k_bin_sort      block = (window, bin): LDS histogram/scan over the LOW bits ("sub"), emits the per-bucket
//                   counts + offsets and the final sorted[] array; all traffic of a block stays inside its bin
constexpr uint32_t PART_TILE = 4096;
constexpr uint32_t PART_MAX_N = 1u << 23;      // idx 23 bits | sign 1 bit | sub 8 bits

__device__ __forceinline__ uint32_t digit_mag(uint32_t e, uint32_t& neg) {   // e != 0: u16 two's complement digit
  if (e == 0x8000u) { neg = 0; return 0x8000u; }                             // +2^15 (only for c = 16)
  int d = (int)(int16_t)(uint16_t)e;
  neg = d < 0;
  return (uint32_t)(d < 0 ? -d : d);
}

// LDS counter increment that returns this lane's slot.  When every active lane of the wave carries the SAME
// key (skewed scalars: all-equal, tiny range, recoding-carry window) the wave issues ONE atomic for all of them
// instead of 64 serialised same-address atomics.  Must be called convergently by the whole wave.
__device__ int g_wave_agg = 1;         // A/B switch (cg1_ctx_set_param "wave_agg")
__device__ __forceinline__ uint32_t lds_ranked_inc(uint32_t* ctr, uint32_t key, bool active) {
  if (!g_wave_agg) return active ? atomicAdd(&ctr[key], 1u) : 0u;
  const unsigned long long amask = __ballot(active);
  if (amask == 0ull) return 0u;
  const int leader = __ffsll((long long)amask) - 1;
  const uint32_t k0 = __shfl(key, leader, 64);
  if (__ballot(active && key != k0) == 0ull) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&ctr[k0], (uint32_t)__popcll(amask));
    base = __shfl(base, leader, 64);
    return base + (uint32_t)__popcll(amask & ((1ull << lane) - 1ull));
  }
  return active ? atomicAdd(&ctr[key], 1u) : 0u;
}

__global__ void __launch_bounds__(256) k_digits(const uint32_t* __restrict__ scalars, const uint8_t* __restrict__ inf_flag,
                                                uint16_t* __restrict__ digits, uint32_t n, int c, int nwin, int rank, int world) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const bool inf = inf_flag[i] != 0;
  DigitIter it; it.c = c;
  load_scalar(scalars, i, it);
  for (int w = 0; w < nwin; ++w) {
    int d = it.next(w);
    if ((w % world) != rank) continue;
    digits[(size_t)(w / world) * n + i] = inf ? (uint16_t)0 : (uint16_t)(d & 0xFFFF);
  }
}

__global__ void __launch_bounds__(256) k_part_count(const uint16_t* __restrict__ digits, uint32_t* __restrict__ block_counts,
                                                    uint32_t n, uint32_t nslices, uint32_t nbins, uint32_t sub_bits) {
  __shared__ uint32_t cnt[128];
  if (threadIdx.x < 128) cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t slice = blockIdx.x, lw = blockIdx.y;
  const uint16_t* dg = digits + (size_t)lw * n;
  for (uint32_t k = 0; k < PART_TILE / 256; ++k) {
    uint32_t i = slice * PART_TILE + k * 256 + threadIdx.x;
    uint32_t e = (i < n) ? (uint32_t)dg[i] : 0u;
    uint32_t neg, key = e ? ((digit_mag(e, neg) - 1u) >> sub_bits) : 0u;
    lds_ranked_inc(cnt, key, e != 0u);
  }
  __syncthreads();
  if (threadIdx.x < nbins) block_counts[((size_t)lw * nbins + threadIdx.x) * nslices + slice] = cnt[threadIdx.x];
}

__global__ void __launch_bounds__(256) k_part_scatter(const uint16_t* __restrict__ digits, const uint32_t* __restrict__ block_base,
                                                      uint32_t* __restrict__ part, uint32_t n, uint32_t nslices, uint32_t nbins,
                                                      uint32_t sub_bits) {
  __shared__ uint32_t cur[128];
  const uint32_t slice = blockIdx.x, lw = blockIdx.y;
  if (threadIdx.x < nbins) cur[threadIdx.x] = block_base[((size_t)lw * nbins + threadIdx.x) * nslices + slice];
  __syncthreads();
  const uint16_t* dg = digits + (size_t)lw * n;
  const uint32_t sub_mask = (1u << sub_bits) - 1u;
  for (uint32_t k = 0; k < PART_TILE / 256; ++k) {
    uint32_t i = slice * PART_TILE + k * 256 + threadIdx.x;
    uint32_t e = (i < n) ? (uint32_t)dg[i] : 0u;
    uint32_t neg = 0, b = e ? (digit_mag(e, neg) - 1u) : 0u;
    uint32_t pos = lds_ranked_inc(cur, b >> sub_bits, e != 0u);
    if (e) part[pos] = i | (neg << 23) | ((b & sub_mask) << 24);
  }
}

__global__ void __launch_bounds__(256) k_bin_sort(const uint32_t* __restrict__ part, const uint32_t* __restrict__ block_base,
                                                  uint32_t* __restrict__ hist, uint32_t* __restrict__ sorted,
                                                  uint32_t nbins_total, uint32_t nslices, uint32_t sub_bits) {
  __shared__ uint32_t cnt[256];
  __shared__ uint32_t cur[256];
  const uint32_t g = blockIdx.x;
  const uint32_t start = block_base[(size_t)g * nslices];
  const uint32_t end = block_base[(size_t)(g + 1) * nslices];     // element [nbins_total*nslices] holds the grand total
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t span = ((end - start + 255u) / 256u) * 256u;       // whole waves iterate together
  for (uint32_t o = threadIdx.x; o < span; o += 256) {
    const bool live = start + o < end;
    lds_ranked_inc(cnt, live ? (part[start + o] >> 24) : 0u, live);
  }
  __syncthreads();
  const uint32_t mine = cnt[threadIdx.x];
  cur[threadIdx.x] = mine;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    uint32_t u = ((int)threadIdx.x >= d) ? cur[threadIdx.x - d] : 0u;
    __syncthreads();
    cur[threadIdx.x] += u;
    __syncthreads();
  }
  const uint32_t excl = cur[threadIdx.x] - mine;
  __syncthreads();
  cur[threadIdx.x] = start + excl;
  if (threadIdx.x < (1u << sub_bits)) hist[((size_t)g << sub_bits) + threadIdx.x] = mine;
  __syncthreads();
  for (uint32_t o = threadIdx.x; o < span; o += 256) {
    const bool live = start + o < end;
    const uint32_t v = live ? part[start + o] : 0u;
    const uint32_t pos = lds_ranked_inc(cur, v >> 24, live);
    if (live) sorted[pos] = (v & 0x7fffffu) | ((v >> 23 & 1u) << 31);
  }
}

// generic in-place exclusive scan of u32 (total written to a[n])
__global__ void __launch_bounds__(256) k_uscan1(uint32_t* __restrict__ a, uint32_t* __restrict__ block_tot, uint32_t n) {
  __shared__ uint32_t sh[256];
  uint32_t base = blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
  uint32_t v[4], local = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { v[k] = (base + k < n) ? a[base + k] : 0u; local += v[k]; }
  sh[threadIdx.x] = local;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    uint32_t u = ((int)threadIdx.x >= d) ? sh[threadIdx.x - d] : 0u;
    __syncthreads();
    sh[threadIdx.x] += u;
    __syncthreads();
  }
  uint32_t excl = sh[threadIdx.x] - local;
#pragma unroll
  for (int k = 0; k < 4; ++k) { if (base + k < n) a[base + k] = excl; excl += v[k]; }
  if (threadIdx.x == 255) block_tot[blockIdx.x] = sh[255];
}
__global__ void __launch_bounds__(256) k_uscan2(uint32_t* __restrict__ block_tot, uint32_t nblocks, uint32_t* __restrict__ a, uint32_t n) {
  __shared__ uint32_t sh[256];
  uint32_t carry = 0;
  for (uint32_t tile = 0; tile < nblocks; tile += 256) {
    uint32_t i = tile + threadIdx.x;
    uint32_t v = (i < nblocks) ? block_tot[i] : 0u;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      uint32_t u = ((int)threadIdx.x >= d) ? sh[threadIdx.x - d] : 0u;
      __syncthreads();
      sh[threadIdx.x] += u;
      __syncthreads();
    }
    if (i < nblocks) block_tot[i] = carry + sh[threadIdx.x] - v;
    uint32_t tot = sh[255];
    __syncthreads();
    carry += tot;
  }
  if (threadIdx.x == 0) a[n] = carry;
}
__global__ void __launch_bounds__(256) k_uscan3(const uint32_t* __restrict__ block_tot, uint32_t* __restrict__ a, uint32_t n) {
  uint32_t base = blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
  uint32_t p = block_tot[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; ++k) if (base + k < n) a[base + k] += p;
}

// ------------------------------------------------------------------ regime B: many independent small MSMs
// A batch of M MSMs (MSM j = terms [offs[j], offs[j+1]) of one concatenated input) -- e.g. the 5*ell+7-term
// final MSMs of 1024 MSMAccumulator.verify() calls (msm_accumulator.py:60-68).  Bucket space is indexed by
// group g = j*nwin + w; with NB <= 256 buckets per group the counting sort of a group lives in one block's
// LDS.  Everything between the sort and the bucket sums (chunking, length ordering, k_accumulate, k_seg_reduce)
// is the same code as regime A, so lanes of one wave work on chunks of equal length from ANY msm/window.
__global__ void __launch_bounds__(256) k_group_count(const uint16_t* __restrict__ digits, const uint32_t* __restrict__ offs,
                                                     uint32_t* __restrict__ hist, uint32_t N, uint32_t NB, uint32_t nwin) {
  __shared__ uint32_t cnt[256];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t j = blockIdx.x, w = blockIdx.y;
  const uint32_t o0 = offs[j], o1 = offs[j + 1];
  const uint16_t* dg = digits + (size_t)w * N;
  const uint32_t span = ((o1 - o0 + 255u) / 256u) * 256u;
  for (uint32_t o = threadIdx.x; o < span; o += 256) {
    const uint32_t e = (o0 + o < o1) ? (uint32_t)dg[o0 + o] : 0u;
    uint32_t neg;
    lds_ranked_inc(cnt, e ? digit_mag(e, neg) - 1u : 0u, e != 0u);
  }
  __syncthreads();
  if (threadIdx.x < NB) hist[((size_t)j * nwin + w) * NB + threadIdx.x] = cnt[threadIdx.x];
}

__global__ void __launch_bounds__(256) k_group_scatter(const uint16_t* __restrict__ digits, const uint32_t* __restrict__ offs,
                                                       const uint32_t* __restrict__ off, uint32_t* __restrict__ sorted,
                                                       uint32_t N, uint32_t NB, uint32_t nwin) {
  __shared__ uint32_t cur[256];
  const uint32_t j = blockIdx.x, w = blockIdx.y;
  if (threadIdx.x < NB) cur[threadIdx.x] = off[((size_t)j * nwin + w) * NB + threadIdx.x];
  __syncthreads();
  const uint32_t o0 = offs[j], o1 = offs[j + 1];
  const uint16_t* dg = digits + (size_t)w * N;
  const uint32_t span = ((o1 - o0 + 255u) / 256u) * 256u;
  for (uint32_t o = threadIdx.x; o < span; o += 256) {
    const uint32_t i = o0 + o;
    const uint32_t e = (i < o1) ? (uint32_t)dg[i] : 0u;
    uint32_t neg = 0;
    const uint32_t b = e ? digit_mag(e, neg) - 1u : 0u;
    const uint32_t pos = lds_ranked_inc(cur, b, e != 0u);
    if (e) sorted[pos] = i | (neg << 31);
  }
}

// one lane per group: S_g = sum_s tot_s + m * sum_s s*run_s over the group's J = NB/m segments
__global__ void __launch_bounds__(256) k_group_reduce(const PointSum* __restrict__ seg_run, const PointSum* __restrict__ seg_tot,
                                                      PointSum* __restrict__ group_sum, uint32_t ngroups, uint32_t J, uint32_t log2m) {
  uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= ngroups) return;
  const PointSum* run = seg_run + (size_t)g * J;
  const PointSum* tot = seg_tot + (size_t)g * J;
  xyzz r = xyzz_identity(), t = xyzz_identity();
  for (uint32_t s = J - 1; s >= 1; --s) { r = xyzz_add(r, load_sum(run + s)); t = xyzz_add(t, r); }   // t = sum s*run_s
  for (uint32_t k = 0; k < log2m; ++k) t = xyzz_dbl(t);
  for (uint32_t s = 0; s < J; ++s) t = xyzz_add(t, load_sum(tot + s));
  store_sum(group_sum + g, t);
}

// one lane per MSM: Horner over its nwin window sums, result as canonical words
__global__ void __launch_bounds__(64) k_msm_horner(const PointSum* __restrict__ group_sum, PointWords* __restrict__ out,
                                                   uint32_t M, uint32_t nwin, uint32_t c) {
  uint32_t j = blockIdx.x * 64 + threadIdx.x;
  if (j >= M) return;
  xyzz acc = xyzz_identity();
  for (int w = (int)nwin - 1; w >= 0; --w) {
    for (uint32_t k = 0; k < c; ++k) acc = xyzz_dbl(acc);
    acc = xyzz_add(acc, load_sum(group_sum + (size_t)j * nwin + w));
  }
  xyzz_words o;
  xyzz_export(acc, o);
  PointWords* dst = out + j;
  for (int cidx = 0; cidx < 4; ++cidx) for (int k = 0; k < 12; ++k) dst->w[cidx][k] = o.w[cidx][k];
  dst->inf = o.inf;
}

// ------------------------------------------------------------------ scan of (count, chunks)
// Chunk length of a bucket with `cnt` entries: L0 normally; a skewed bucket (e.g. the reference's [beta]*ell
// all-equal scalars, same_perm.py:54-55, or the recoding carry of a top window) is cut into at most
// MAX_CHUNKS_PER_BUCKET pieces whose partial sums are then combined by a block-wide tree (k_heavy_combine)
// instead of serialising one lane.
constexpr uint32_t MAX_CHUNKS_PER_BUCKET = 4096;
constexpr uint32_t HEAVY_MIN_CHUNKS = 17;       // buckets with >= this many chunks go through k_heavy_combine;
                                                // 2..16 chunks are folded by one lane in k_bucket_fold
__device__ __forceinline__ uint32_t chunk_len(uint32_t cnt, uint32_t L0) {
  uint32_t s = (cnt + MAX_CHUNKS_PER_BUCKET - 1) / MAX_CHUNKS_PER_BUCKET;
  return s > L0 ? s : L0;
}
__device__ __forceinline__ uint32_t chunk_count(uint32_t cnt, uint32_t L0) {
  if (cnt == 0) return 0;
  uint32_t L = chunk_len(cnt, L0);
  return (cnt + L - 1) / L;
}

// phase 1: per-block exclusive scan, block totals out
__global__ void __launch_bounds__(256) k_scan1(const uint32_t* __restrict__ hist, uint32_t* __restrict__ off, uint32_t* __restrict__ choff,
                                               uint2* __restrict__ block_tot, uint32_t nb_total, uint32_t L0) {
  __shared__ uint2 sh[256];
  uint32_t base = blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
  uint32_t cnt[4], ch[4];
  uint2 local = make_uint2(0, 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    cnt[k] = (base + k < nb_total) ? hist[base + k] : 0u;
    ch[k] = chunk_count(cnt[k], L0);
    local.x += cnt[k]; local.y += ch[k];
  }
  sh[threadIdx.x] = local;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {          // Hillis-Steele inclusive scan over 256 partials
    uint2 v = make_uint2(0, 0);
    if ((int)threadIdx.x >= d) v = sh[threadIdx.x - d];
    __syncthreads();
    sh[threadIdx.x].x += v.x; sh[threadIdx.x].y += v.y;
    __syncthreads();
  }
  uint2 excl = make_uint2(sh[threadIdx.x].x - local.x, sh[threadIdx.x].y - local.y);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (base + k < nb_total) { off[base + k] = excl.x; choff[base + k] = excl.y; }
    excl.x += cnt[k]; excl.y += ch[k];
  }
  if (threadIdx.x == 255) block_tot[blockIdx.x] = sh[255];
}
// phase 2: one block turns block totals into exclusive block prefixes (serial over tiles of 256)
__global__ void __launch_bounds__(256) k_scan2(uint2* __restrict__ block_tot, uint32_t nblocks, uint32_t* __restrict__ off,
                                               uint32_t* __restrict__ choff, uint32_t nb_total) {
  __shared__ uint2 sh[256];
  uint2 carry = make_uint2(0, 0);
  for (uint32_t tile = 0; tile < nblocks; tile += 256) {
    uint32_t i = tile + threadIdx.x;
    uint2 v = (i < nblocks) ? block_tot[i] : make_uint2(0, 0);
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      uint2 u = make_uint2(0, 0);
      if ((int)threadIdx.x >= d) u = sh[threadIdx.x - d];
      __syncthreads();
      sh[threadIdx.x].x += u.x; sh[threadIdx.x].y += u.y;
      __syncthreads();
    }
    if (i < nblocks) block_tot[i] = make_uint2(carry.x + sh[threadIdx.x].x - v.x, carry.y + sh[threadIdx.x].y - v.y);
    uint2 tot = sh[255];
    __syncthreads();
    carry.x += tot.x; carry.y += tot.y;
  }
  if (threadIdx.x == 0) { off[nb_total] = carry.x; choff[nb_total] = carry.y; }
}
__global__ void __launch_bounds__(256) k_scan3(const uint2* __restrict__ block_tot, uint32_t* __restrict__ off,
                                               uint32_t* __restrict__ choff, uint32_t nb_total) {
  uint32_t base = blockIdx.x * SCAN_ITEMS + threadIdx.x * 4;
  uint2 p = block_tot[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 4; ++k) if (base + k < nb_total) { off[base + k] += p.x; choff[base + k] += p.y; }
}

constexpr uint32_t LEN_BINS = 256;      // chunk-length keys: min(len, 255)
__device__ __forceinline__ uint32_t len_key(uint32_t len) { return len < LEN_BINS - 1 ? len : LEN_BINS - 1; }

__global__ void __launch_bounds__(256) k_chunk_desc(const uint32_t* __restrict__ off, const uint32_t* __restrict__ choff,
                                                    uint2* __restrict__ desc, uint32_t* __restrict__ len_hist,
                                                    uint32_t* __restrict__ heavy /* [0] = count, then bucket ids */,
                                                    uint32_t heavy_cap, uint32_t nb_total, uint32_t L0) {
  __shared__ uint32_t sh[LEN_BINS];
  sh[threadIdx.x] = 0;
  __syncthreads();
  uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b < nb_total) {
    uint32_t start = off[b], cnt = off[b + 1] - start;
    if (cnt) {
      uint32_t L = chunk_len(cnt, L0), nch = (cnt + L - 1) / L, o = choff[b];
      for (uint32_t k = 0; k < nch; ++k) {
        uint32_t s = k * L, len = (cnt - s < L) ? (cnt - s) : L;
        desc[o + k] = make_uint2(start + s, len);
        atomicAdd(&sh[len_key(len)], 1u);
      }
      if (nch >= HEAVY_MIN_CHUNKS) {
        uint32_t slot = atomicAdd(&heavy[0], 1u);
        if (slot < heavy_cap) heavy[1 + slot] = b;
      }
    }
  }
  __syncthreads();
  uint32_t v = sh[threadIdx.x];
  if (v) atomicAdd(&len_hist[threadIdx.x], v);
}

// one block: len_cursor[k] = number of chunks with a LONGER key (descending order => longest chunks first)
__global__ void __launch_bounds__(256) k_len_scan(const uint32_t* __restrict__ len_hist, uint32_t* __restrict__ len_cursor) {
  __shared__ uint32_t sh[LEN_BINS];
  uint32_t rev = LEN_BINS - 1 - threadIdx.x;          // thread i handles key 255-i
  uint32_t v = len_hist[rev];
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    uint32_t u = ((int)threadIdx.x >= d) ? sh[threadIdx.x - d] : 0u;
    __syncthreads();
    sh[threadIdx.x] += u;
    __syncthreads();
  }
  len_cursor[rev] = sh[threadIdx.x] - v;
}

// order[] = chunk ids sorted by descending length key (stable enough: order inside a key is arbitrary)
__global__ void __launch_bounds__(256) k_order(const uint2* __restrict__ desc, const uint32_t* __restrict__ total_chunks,
                                               uint32_t* __restrict__ len_cursor, uint32_t* __restrict__ order) {
  __shared__ uint32_t cnt[LEN_BINS];
  __shared__ uint32_t base[LEN_BINS];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  uint32_t t = blockIdx.x * 256 + threadIdx.x;
  bool live = t < *total_chunks;
  uint32_t key = 0, local = 0;
  if (live) { key = len_key(desc[t].y); local = atomicAdd(&cnt[key], 1u); }
  __syncthreads();
  uint32_t c = cnt[threadIdx.x];
  if (c) base[threadIdx.x] = atomicAdd(&len_cursor[threadIdx.x], c);
  __syncthreads();
  if (live) order[base[key] + local] = t;
}

// ------------------------------------------------------------------ k_accumulate (dominant kernel)
__global__ void __launch_bounds__(256) k_accumulate(const uint2* __restrict__ desc, const uint32_t* __restrict__ total_chunks,
                                                    const uint32_t* __restrict__ order, const uint32_t* __restrict__ sorted,
                                                    const PreparedPoint* __restrict__ pts, PointSum* __restrict__ sums) {
  uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= *total_chunks) return;
  const uint32_t t = order[g];              // chunks in descending length: lanes of a wave finish together
  const uint2 d = desc[t];
  const uint32_t* ent = sorted + d.x;
  xyzz acc = xyzz_identity();
  uint32_t e = ent[0];
  fp x, y; uint32_t flags;
  load_affine(pts + (e & 0x7fffffffu), x, y, flags);
  for (uint32_t j = 0; j < d.y; ++j) {
    // prefetch the next entry's point while this one is being added
    uint32_t en = ent[(j + 1 < d.y) ? j + 1 : j];
    fp xn, yn;
    load_affine(pts + (en & 0x7fffffffu), xn, yn, flags);
    if (e >> 31) y = fp_neg<3>(y);
    acc = xyzz_madd(acc, x, y);
    e = en; x = xn; y = yn;
  }
  store_sum(sums + t, acc);
}

// ------------------------------------------------------------------ k_heavy_combine
// Blocks stride over the heavy-bucket list; one block adds ALL chunk sums of its bucket (<= 4096):
// <= 16 serial adds per lane, then wave shuffles, then LDS.  The total replaces the bucket's first chunk sum and
// the bucket is flagged so k_seg_reduce reads only that slot.
__device__ __forceinline__ xyzz shfl_down_xyzz(const xyzz& a, int delta);
__global__ void __launch_bounds__(256) k_heavy_combine(const uint32_t* __restrict__ heavy, uint32_t heavy_cap,
                                                       const uint32_t* __restrict__ choff, PointSum* __restrict__ sums,
                                                       uint8_t* __restrict__ combined) {
  __shared__ PointSum sh[4];
  uint32_t nheavy = heavy[0];
  if (nheavy > heavy_cap) nheavy = heavy_cap;
  for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
    const uint32_t b = heavy[1 + h];
    const uint32_t c0 = choff[b], c1 = choff[b + 1];
    xyzz acc = xyzz_identity();
    for (uint32_t k = c0 + threadIdx.x; k < c1; k += 256) acc = xyzz_add(acc, load_sum(sums + k));
    __syncthreads();                     // every chunk sum has been read before slot c0 is overwritten
    for (int delta = 32; delta >= 1; delta >>= 1) {
      xyzz o = shfl_down_xyzz(acc, delta);
      if ((threadIdx.x & 63) < (uint32_t)delta) acc = xyzz_add(acc, o);
    }
    if ((threadIdx.x & 63) == 0) store_sum(&sh[threadIdx.x >> 6], acc);
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int k = 1; k < 4; ++k) acc = xyzz_add(acc, load_sum(&sh[k]));
      store_sum(sums + c0, acc);
      combined[b] = 1;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ k_bucket_fold
// One lane per bucket: buckets whose entries were cut into 2..16 chunks (a window-sharded rank owns few buckets,
// so chunks are kept short for parallelism in k_accumulate) get their chunk sums added serially into the first
// slot.  Buckets with more chunks were already handled by k_heavy_combine; single-chunk buckets are untouched.
__global__ void __launch_bounds__(256) k_bucket_fold(const uint32_t* __restrict__ choff, PointSum* __restrict__ sums,
                                                     uint8_t* __restrict__ combined, uint32_t nb_total) {
  uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nb_total) return;
  const uint32_t c0 = choff[b], c1 = choff[b + 1];
  if (c1 - c0 < 2u || c1 - c0 >= HEAVY_MIN_CHUNKS) return;
  xyzz acc = load_sum(sums + c0);
  for (uint32_t k = c0 + 1; k < c1; ++k) acc = xyzz_add(acc, load_sum(sums + k));
  store_sum(sums + c0, acc);
  combined[b] = 1;
}

// ------------------------------------------------------------------ k_seg_reduce
// One lane per segment of `m` consecutive buckets of one window.  Bucket b of the window carries digit
// value b+1.  Emits run_j = sum_t B_{jm+t}, tot_j = sum_t (t+1) B_{jm+t}  (t = 0..m-1).
__global__ void __launch_bounds__(256) k_seg_reduce(const uint32_t* __restrict__ choff, const PointSum* __restrict__ sums,
                                                    const uint8_t* __restrict__ combined,
                                                    PointSum* __restrict__ seg_run, PointSum* __restrict__ seg_tot,
                                                    uint32_t nseg_total, uint32_t m) {
  uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= nseg_total) return;
  xyzz run = xyzz_identity(), tot = xyzz_identity();
  for (int t = (int)m - 1; t >= 0; --t) {
    uint32_t b = s * m + (uint32_t)t;           // segments tile the flat (window, bucket) array
    uint32_t c0 = choff[b], c1 = choff[b + 1];
    if (combined[b]) c1 = c0 + 1;        // k_heavy_combine already folded all chunks into the first slot
    for (uint32_t k = c0; k < c1; ++k) run = xyzz_add(run, load_sum(sums + k));
    tot = xyzz_add(tot, run);
  }
  store_sum(seg_run + s, run);
  store_sum(seg_tot + s, tot);
}

// ------------------------------------------------------------------ k_bit_tree
__device__ __forceinline__ xyzz shfl_down_xyzz(const xyzz& a, int delta) {
  xyzz r;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    r.X.l[i] = __shfl_down(a.X.l[i], delta, 64);
    r.Y.l[i] = __shfl_down(a.Y.l[i], delta, 64);
    r.ZZ.l[i] = __shfl_down(a.ZZ.l[i], delta, 64);
    r.ZZZ.l[i] = __shfl_down(a.ZZZ.l[i], delta, 64);
  }
  r.inf = __shfl_down(a.inf, delta, 64);
  return r;
}

// stage 1: grid = (nitems, nlw, S).  item 0: T = sum_j seg_tot[j];  item 1+b: Y_b = sum_{j: bit b of j} seg_run[j].
// Only the SELECTED j are enumerated (all J for item 0, the J/2 with bit b set otherwise) so no lane idles in
// the serial part; block z takes BT_ELEMS consecutive selected elements (8 per lane).  The cost model that
// shaped this: one wave-level EC add step is ~25 us and the chip runs 2048 of them at once, so total
// wave-steps = waves x (elements per lane + 6 shuffle levels + 2) must be kept small, not just the depth.
constexpr uint32_t BT_ELEMS = 2048;
__global__ void __launch_bounds__(256) k_bit_tree(const PointSum* __restrict__ seg_run, const PointSum* __restrict__ seg_tot,
                                                  PointSum* __restrict__ partial, uint32_t J) {
  __shared__ PointSum sh[4];
  const uint32_t item = blockIdx.x, lw = blockIdx.y, S = gridDim.z, z = blockIdx.z;
  const PointSum* src = (item == 0 ? seg_tot : seg_run) + (size_t)lw * J;
  const uint32_t count = (item == 0) ? J : (J >> 1);
  const uint32_t e0 = z * BT_ELEMS, e1 = (e0 + BT_ELEMS < count) ? e0 + BT_ELEMS : count;
  xyzz acc = xyzz_identity();
  const uint32_t b = item - 1;                   // bit index for item >= 1
  for (uint32_t e = e0 + threadIdx.x; e < e1; e += 256) {
    // e-th index with bit b set: insert a 1 at bit position b
    uint32_t j = (item == 0) ? e : ((((e >> b) << 1) | 1u) << b) | (e & ((1u << b) - 1u));
    acc = xyzz_add(acc, load_sum(src + j));
  }
  for (int delta = 32; delta >= 1; delta >>= 1) {
    xyzz o = shfl_down_xyzz(acc, delta);
    if ((threadIdx.x & 63) < (uint32_t)delta) acc = xyzz_add(acc, o);
  }
  if ((threadIdx.x & 63) == 0) store_sum(&sh[threadIdx.x >> 6], acc);
  __syncthreads();
  if (threadIdx.x < 64) {
    acc = (threadIdx.x < 4) ? load_sum(&sh[threadIdx.x]) : xyzz_identity();
    for (int delta = 2; delta >= 1; delta >>= 1) {
      xyzz o = shfl_down_xyzz(acc, delta);
      if (threadIdx.x < (uint32_t)delta) acc = xyzz_add(acc, o);
    }
    if (threadIdx.x == 0) store_sum(partial + ((size_t)lw * gridDim.x + item) * S + z, acc);
  }
}

// stage 2: one wave per (window, item): shuffle-tree over the S <= 64 slice partials, export canonical words
__global__ void __launch_bounds__(64) k_bit_tree_final(const PointSum* __restrict__ partial, PointWords* __restrict__ out, uint32_t S) {
  const size_t idx = blockIdx.x;
  xyzz acc = (threadIdx.x < S) ? load_sum(partial + idx * S + threadIdx.x) : xyzz_identity();
  for (int delta = 32; delta >= 1; delta >>= 1) {
    if ((uint32_t)delta >= S) continue;      // wave-uniform
    xyzz o = shfl_down_xyzz(acc, delta);
    if (threadIdx.x < (uint32_t)delta) acc = xyzz_add(acc, o);
  }
  if (threadIdx.x == 0) {
    xyzz_words o;
    xyzz_export(acc, o);
    PointWords* dst = out + idx;
    for (int cidx = 0; cidx < 4; ++cidx) for (int k = 0; k < 12; ++k) dst->w[cidx][k] = o.w[cidx][k];
    dst->inf = o.inf;
  }
}

// ------------------------------------------------------------------ 2-D bucket reduction (regime A default)
// S_w = sum_b (b+1) B_b over the window's 2^(c-1) buckets, b = h * 2^lb + l:
//     S_w = T0 + 2^lb * sum_h h A_h + sum_l l C_l,   A_h = sum_l B_{h,l} (row sums),  C_l = sum_h B_{h,l} (column sums),
//     T0 = sum_h A_h.
// k_rowcol forms all row and column sums in ONE launch (2 EC adds per bucket; blocks [0, nrow_blocks) take rows,
// the rest columns; <= 8 serial adds per lane, then a shuffle tree inside 32 / 16 lanes).  k_small_tree then
// turns the 2^hb row sums and 2^lb column sums of a window into 1 + hb + lb points (plain sum + one masked sum per
// index bit) whose power-of-two weights the host Horner applies.  Versus k_seg_reduce + k_bit_tree this halves
// the wave-level EC-add steps (47 K -> ~20 K at c = 16) and shortens the dependent chain (~27 -> ~22 steps).
__device__ __forceinline__ xyzz bucket_sum(const uint32_t* __restrict__ choff, const PointSum* __restrict__ sums,
                                           const uint8_t* __restrict__ combined, uint32_t b) {
  uint32_t c0 = choff[b], c1 = choff[b + 1];
  if (combined[b]) c1 = c0 + 1;
  xyzz acc = xyzz_identity();
  for (uint32_t k = c0; k < c1; ++k) acc = xyzz_add(acc, load_sum(sums + k));
  return acc;
}

__global__ void __launch_bounds__(256, 2) k_rowcol(const uint32_t* __restrict__ choff, const PointSum* __restrict__ sums,
                                                const uint8_t* __restrict__ combined, PointSum* __restrict__ rowsum,
                                                PointSum* __restrict__ colsum, uint32_t nlw, uint32_t hb, uint32_t lb,
                                                uint32_t nrow_blocks) {
  const uint32_t R = 1u << hb, Cn = 1u << lb;
  if (blockIdx.x < nrow_blocks) {
    const uint32_t lpr = Cn < 32u ? Cn : 32u, serial = Cn / lpr;
    const uint32_t gr = blockIdx.x * (256u / lpr) + threadIdx.x / lpr;     // global row = lw * R + h
    const uint32_t part = threadIdx.x % lpr;
    const bool live = gr < nlw * R;
    xyzz acc = xyzz_identity();
    if (live)
      for (uint32_t t = 0; t < serial; ++t) acc = xyzz_add(acc, bucket_sum(choff, sums, combined, gr * Cn + part * serial + t));
    for (uint32_t delta = lpr >> 1; delta >= 1; delta >>= 1) {
      xyzz o = shfl_down_xyzz(acc, (int)delta);
      if (part < delta) acc = xyzz_add(acc, o);
    }
    if (live && part == 0) store_sum(rowsum + gr, acc);
  } else {
    const uint32_t lpc = R < 16u ? R : 16u, serial = R / lpc;
    const uint32_t gc = (blockIdx.x - nrow_blocks) * (256u / lpc) + threadIdx.x / lpc;   // global column = lw * Cn + l
    const uint32_t part = threadIdx.x % lpc;
    const bool live = gc < nlw * Cn;
    const uint32_t lw = gc / Cn, l = gc % Cn;
    xyzz acc = xyzz_identity();
    if (live)
      for (uint32_t t = 0; t < serial; ++t) acc = xyzz_add(acc, bucket_sum(choff, sums, combined, (lw * R + part * serial + t) * Cn + l));
    for (uint32_t delta = lpc >> 1; delta >= 1; delta >>= 1) {
      xyzz o = shfl_down_xyzz(acc, (int)delta);
      if (part < delta) acc = xyzz_add(acc, o);
    }
    if (live && part == 0) store_sum(colsum + gc, acc);
  }
}

// grid = (1 + hb + lb, nlw), 256 threads.  item 0: T0 = sum_h A_h; item 1+k (k < hb): sum of A_h with bit k of h set;
// item 1+hb+k (k < lb): sum of C_l with bit k of l set.  Requires 2^hb, 2^lb <= 256.  Emits canonical words.
__global__ void __launch_bounds__(256) k_small_tree(const PointSum* __restrict__ rowsum, const PointSum* __restrict__ colsum,
                                                    PointWords* __restrict__ out, uint32_t hb, uint32_t lb) {
  __shared__ PointSum sh[4];
  const uint32_t item = blockIdx.x, lw = blockIdx.y;
  const bool on_rows = item <= hb;
  const uint32_t J = on_rows ? (1u << hb) : (1u << lb);
  const PointSum* src = on_rows ? rowsum + (size_t)lw * J : colsum + (size_t)lw * J;
  const uint32_t bit = on_rows ? item - 1u : item - 1u - hb;           // unused for item 0
  xyzz acc = xyzz_identity();
  if (threadIdx.x < J && (item == 0 || ((threadIdx.x >> bit) & 1u))) acc = load_sum(src + threadIdx.x);
  for (int delta = 32; delta >= 1; delta >>= 1) {
    xyzz o = shfl_down_xyzz(acc, delta);
    if ((threadIdx.x & 63) < (uint32_t)delta) acc = xyzz_add(acc, o);
  }
  if ((threadIdx.x & 63) == 0) store_sum(&sh[threadIdx.x >> 6], acc);
  __syncthreads();
  if (threadIdx.x < 64) {
    acc = (threadIdx.x < 4) ? load_sum(&sh[threadIdx.x]) : xyzz_identity();
    for (int delta = 2; delta >= 1; delta >>= 1) {
      xyzz o = shfl_down_xyzz(acc, delta);
      if (threadIdx.x < (uint32_t)delta) acc = xyzz_add(acc, o);
    }
    if (threadIdx.x == 0) {
      xyzz_words o;
      xyzz_export(acc, o);
      PointWords* dst = out + (size_t)lw * gridDim.x + item;
      for (int cidx = 0; cidx < 4; ++cidx) for (int k = 0; k < 12; ++k) dst->w[cidx][k] = o.w[cidx][k];
      dst->inf = o.inf;
    }
  }
}

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

// ------------------------------------------------------------------ batched 48-byte G1 decompression
// One lane per point: parse the ZCash-format encoding (util.py:35-36 -> G1Point.from_compressed_bytes[_unchecked]),
// y = sqrt(x^3 + 4) by exponentiation, sign select, optional subgroup test  [z^2]P == phi(P) + P.
// out: affine96 (zeros = identity), status: 0 ok, CG1_ERR_ENCODING / _NOT_ON_CURVE / _NOT_IN_SUBGROUP.
__global__ void __launch_bounds__(128) k_batch_decompress(const uint8_t* __restrict__ in48, uint32_t* __restrict__ out_raw,
                                                          uint8_t* __restrict__ status, uint32_t n, int check_subgroup) {
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
  if (!compressed || (infinity && largest)) { status[i] = CG1_ERR_ENCODING; return; }
  if (infinity) {
    uint32_t any = 0;
    for (int j = 0; j < 12; ++j) any |= w[j];
    status[i] = any ? CG1_ERR_ENCODING : CG1_OK;
    return;
  }
  bool lt = false, decided = false;             // x < p ?
  for (int j = 11; j >= 0 && !decided; --j) if (w[j] != W_P[j]) { lt = w[j] < W_P[j]; decided = true; }
  if (!lt) { status[i] = CG1_ERR_ENCODING; return; }
  const fp x = fp_to_mont(fp_from_words(w));
  fp four = fp_one(); four = fp_dbl(fp_dbl(four));
  const fp rhs = fp_norm(fp_add(fp_mul(fp_sqr(x), x), four));
  fp y = fp_sqrt_candidate(rhs);
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
  if (check_subgroup) {
    constexpr uint32_t bt[NL] = {D_BETA[0], D_BETA[1], D_BETA[2], D_BETA[3], D_BETA[4], D_BETA[5], D_BETA[6], D_BETA[7], D_BETA[8], D_BETA[9], D_BETA[10], D_BETA[11], D_BETA[12], D_BETA[13]};
    fp beta; for (int k = 0; k < NL; ++k) beta.l[k] = bt[k];
    xyzz acc = xyzz_identity();
    for (int bit = 127; bit >= 0; --bit) {       // [z^2] P
      acc = xyzz_dbl(acc);
      if ((H_ZSQ[bit >> 6] >> (bit & 63)) & 1ull) acc = xyzz_madd(acc, x, y);
    }
    const fp yneg = fp_neg<3>(y);
    acc = xyzz_madd(acc, x, yneg);               // - P
    acc = xyzz_madd(acc, fp_mul(x, beta), yneg); // - phi(P)
    if (!acc.inf) { status[i] = CG1_ERR_NOT_IN_SUBGROUP; return; }
  }
  for (int k = 0; k < 12; ++k) { dst[k] = w[k]; dst[12 + k] = yw[k]; }
  status[i] = CG1_OK;
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

// ------------------------------------------------------------------ host-side context
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  snprintf(ctx->err, sizeof ctx->err, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); return CG1_ERR_HIP; } } while (0)

struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  char err[256] = {0};
  // capacity
  size_t cap_n = 0, cap_nb = 0, cap_chunks = 0, cap_entries = 0, cap_out = 0;
  PreparedPoint* d_pts = nullptr;
  uint8_t* d_flags = nullptr;
  uint32_t *d_hist = nullptr, *d_off = nullptr, *d_choff = nullptr, *d_sorted = nullptr;
  uint2 *d_blocktot = nullptr, *d_desc = nullptr;
  uint32_t *d_order = nullptr, *d_lenhist = nullptr;      // [2*LEN_BINS]: histogram, cursor
  PointSum* d_partial = nullptr; size_t cap_partial = 0;
  uint16_t* d_digits = nullptr; uint32_t* d_part = nullptr; uint32_t* d_blockcnt = nullptr; uint32_t* d_ublocktot = nullptr;
  size_t cap_digits = 0, cap_part = 0, cap_blockcnt = 0;
  int use_partition_sort = 1;
  int reduce_2d = 1;                    // 1: k_rowcol + k_small_tree; 0: k_seg_reduce + k_bit_tree (A/B switch)
  uint32_t* d_heavy = nullptr; size_t cap_heavy = 0;         // [0] count, then heavy bucket ids
  uint8_t* d_combined = nullptr; size_t cap_combined = 0;
  uint32_t* d_boffs = nullptr; size_t cap_boffs = 0;          // regime B: MSM offsets, group sums, per-MSM results
  PointSum* d_gsum = nullptr; size_t cap_gsum = 0;
  PointWords* d_bout = nullptr; PointWords* h_bout = nullptr; size_t cap_bout = 0;
  PointSum *d_sums = nullptr, *d_segrun = nullptr, *d_segtot = nullptr;
  PointWords* d_out = nullptr;
  PointWords* h_out = nullptr;          // pinned
  // staging for host-pointer entry points
  void* d_stage_pts = nullptr; void* d_stage_sc = nullptr; size_t cap_stage = 0;
  // timing
  hipEvent_t ev[CG1_NPHASE + 1];
  float phase_ms[CG1_NPHASE] = {0};
  float host_tail_ms = 0;
  float host_ms[4] = {0, 0, 0, 0};      // enqueue, wait-for-GPU, event readout, Horner tail
  int profile = 1;                      // read the per-phase hipEvents after each call
  uint32_t last_chunks = 0, last_entries = 0;
  int last_c = 0;
  uint32_t L0 = 8;                      // MINIMUM chunk length; the per-call length grows with the entry count
  uint32_t seg_m = 4;
};

static void free_bufs(Ctx* c) {
  auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
  F(c->d_pts); F(c->d_flags); F(c->d_hist); F(c->d_off); F(c->d_choff); F(c->d_sorted); F(c->d_blocktot); F(c->d_desc);
  F(c->d_sums); F(c->d_segrun); F(c->d_segtot); F(c->d_out); F(c->d_order); F(c->d_lenhist); F(c->d_partial);
  c->cap_partial = 0;
  F(c->d_digits); F(c->d_part); F(c->d_blockcnt); F(c->d_ublocktot); F(c->d_boffs); F(c->d_gsum); F(c->d_bout);
  if (c->h_bout) { (void)hipHostFree(c->h_bout); c->h_bout = nullptr; }
  c->cap_boffs = c->cap_gsum = c->cap_bout = 0;
  c->cap_digits = c->cap_part = c->cap_blockcnt = 0;
  if (c->h_out) { (void)hipHostFree(c->h_out); c->h_out = nullptr; }
  c->cap_n = c->cap_nb = c->cap_chunks = c->cap_entries = c->cap_out = 0;
}

static int ensure(Ctx* ctx, size_t n, size_t nb_total, size_t nlw, size_t nitems, uint32_t L) {
  size_t entries = n * nlw;
  size_t chunks = nb_total + entries / L + 1;
  if (n > ctx->cap_n) {
    if (ctx->d_pts) (void)hipFree(ctx->d_pts);
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    HIPCHK(hipMalloc(&ctx->d_pts, n * sizeof(PreparedPoint)));
    HIPCHK(hipMalloc(&ctx->d_flags, n + 16));
    ctx->cap_n = n;
  }
  if (nb_total > ctx->cap_nb) {
    auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    F(ctx->d_hist); F(ctx->d_off); F(ctx->d_choff); F(ctx->d_blocktot); F(ctx->d_segrun); F(ctx->d_segtot);
    HIPCHK(hipMalloc(&ctx->d_hist, nb_total * 4));
    HIPCHK(hipMalloc(&ctx->d_off, (nb_total + 1) * 4));
    HIPCHK(hipMalloc(&ctx->d_choff, (nb_total + 1) * 4));
    HIPCHK(hipMalloc(&ctx->d_blocktot, (nb_total / SCAN_ITEMS + 2) * sizeof(uint2)));
    HIPCHK(hipMalloc(&ctx->d_segrun, nb_total * sizeof(PointSum)));   // >= nb_total / m segments
    HIPCHK(hipMalloc(&ctx->d_segtot, nb_total * sizeof(PointSum)));
    ctx->cap_nb = nb_total;
  }
  if (entries > ctx->cap_entries) {
    if (ctx->d_sorted) (void)hipFree(ctx->d_sorted);
    HIPCHK(hipMalloc(&ctx->d_sorted, (entries + 1) * 4));
    ctx->cap_entries = entries;
  }
  if (chunks > ctx->cap_chunks) {
    if (ctx->d_desc) (void)hipFree(ctx->d_desc);
    if (ctx->d_sums) (void)hipFree(ctx->d_sums);
    if (ctx->d_order) (void)hipFree(ctx->d_order);
    HIPCHK(hipMalloc(&ctx->d_order, chunks * 4));
    HIPCHK(hipMalloc(&ctx->d_desc, chunks * sizeof(uint2)));
    HIPCHK(hipMalloc(&ctx->d_sums, chunks * sizeof(PointSum)));
    ctx->cap_chunks = chunks;
  }
  if (!ctx->d_lenhist) HIPCHK(hipMalloc(&ctx->d_lenhist, 2 * LEN_BINS * 4));
  {
    const size_t hcap = entries / ((size_t)L * (HEAVY_MIN_CHUNKS - 1)) + 2;     // a heavy bucket holds > (MIN-1)*L entries
    if (hcap > ctx->cap_heavy) {
      if (ctx->d_heavy) (void)hipFree(ctx->d_heavy);
      HIPCHK(hipMalloc(&ctx->d_heavy, (hcap + 1) * 4));
      ctx->cap_heavy = hcap;
    }
    if (nb_total > ctx->cap_combined) {
      if (ctx->d_combined) (void)hipFree(ctx->d_combined);
      HIPCHK(hipMalloc(&ctx->d_combined, nb_total));
      ctx->cap_combined = nb_total;
    }
  }
  if (ctx->use_partition_sort && n <= PART_MAX_N) {
    const size_t nslices = (n + PART_TILE - 1) / PART_TILE;
    const size_t nbc = nlw * 128 * nslices + 1;            // nbins <= 128
    if (entries > ctx->cap_digits) {
      if (ctx->d_digits) (void)hipFree(ctx->d_digits);
      HIPCHK(hipMalloc(&ctx->d_digits, entries * 2 + 16));
      ctx->cap_digits = entries;
    }
    if (entries > ctx->cap_part) {
      if (ctx->d_part) (void)hipFree(ctx->d_part);
      HIPCHK(hipMalloc(&ctx->d_part, (entries + 1) * 4));
      ctx->cap_part = entries;
    }
    if (nbc > ctx->cap_blockcnt) {
      if (ctx->d_blockcnt) (void)hipFree(ctx->d_blockcnt);
      if (ctx->d_ublocktot) (void)hipFree(ctx->d_ublocktot);
      HIPCHK(hipMalloc(&ctx->d_blockcnt, nbc * 4));
      HIPCHK(hipMalloc(&ctx->d_ublocktot, (nbc / SCAN_ITEMS + 2) * 4));
      ctx->cap_blockcnt = nbc;
    }
  }
  size_t nout = nlw * nitems;
  if (nout * 64 > ctx->cap_partial) {
    if (ctx->d_partial) (void)hipFree(ctx->d_partial);
    HIPCHK(hipMalloc(&ctx->d_partial, nout * 64 * sizeof(PointSum)));
    ctx->cap_partial = nout * 64;
  }
  if (nout > ctx->cap_out) {
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    if (ctx->h_out) (void)hipHostFree(ctx->h_out);
    HIPCHK(hipMalloc(&ctx->d_out, nout * sizeof(PointWords)));
    HIPCHK(hipHostMalloc(&ctx->h_out, nout * sizeof(PointWords)));
    ctx->cap_out = nout;
  }
  return CG1_OK;
}

static cg1h::fe fe_from_words12(const uint32_t w[12]) {
  uint64_t v[6];
  for (int i = 0; i < 6; ++i) v[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
  return cg1h::fe_from_std(v);
}
static cg1h::jac jac_from_words(const PointWords& p) {
  if (p.inf) return cg1h::jac_identity();
  return cg1h::jac_from_xyzz(fe_from_words12(p.w[0]), fe_from_words12(p.w[1]), fe_from_words12(p.w[2]), fe_from_words12(p.w[3]));
}

int pick_window(size_t n) {
  // Only widths whose TOP window still holds >= min(c-1, 7) scalar bits (255 = (nwin-1)*c + t): with t = 2..3 all
  // n terms of that window fall into <= 8 buckets.  Thresholds from tools/gpu_window_sweep.py on MI355X.
  if (n <= 128) return 4;        // t = 3
  if (n <= 8192) return 8;       // t = 7
  return 16;                     // t = 15
}

int msm_device(Ctx* ctx, const void* d_points96, const void* d_scalars32, size_t n, int c, int rank, int world, cg1h::jac& result) {
  result = cg1h::jac_identity();
  if (n == 0) return CG1_OK;
  if (n >= (1ull << 31)) { snprintf(ctx->err, sizeof ctx->err, "n too large"); return CG1_ERR_ARG; }
  if (world < 1 || rank < 0 || rank >= world) { snprintf(ctx->err, sizeof ctx->err, "bad window shard %d/%d", rank, world); return CG1_ERR_ARG; }
  if (c <= 0) c = pick_window(n);
  if (c < 4 || c > 16) { snprintf(ctx->err, sizeof ctx->err, "window width %d out of range [4,16]", c); return CG1_ERR_ARG; }
  HIPCHK(hipSetDevice(ctx->device));
  const int nwin = 255 / c + 1;
  const int nlw = (nwin - rank + world - 1) / world;           // windows w = rank, rank+world, ...
  if (nlw <= 0) return CG1_OK;
  const uint32_t NB = 1u << (c - 1);
  const uint32_t m = std::min<uint32_t>(ctx->seg_m, NB);
  const uint32_t J = NB / m;                                   // segments per window
  int nbits = 0; while ((1u << nbits) < J) ++nbits;
  const uint32_t bb = (uint32_t)c - 1u, lb2 = (bb + 1u) / 2u, hb2 = bb - lb2;       // 2-D split of the bucket index
  const bool use2d = ctx->reduce_2d != 0;
  const uint32_t nitems = use2d ? 1u + hb2 + lb2 : 1u + (uint32_t)nbits;
  const size_t nb_total = (size_t)nlw * NB;
  // chunk length: grows with the total entry count so that k_accumulate keeps >= 2^18 lanes busy without flooding the
  // reduce phases with chunk sums (64 at 2^20 terms x 16 windows, 512 at 2^23) and shrinks to the minimum (8) for
  // small inputs, where the dependent madd chain of one chunk IS the critical path.  Buckets cut into several chunks
  // (window-sharded ranks, skew, thin top windows) are re-joined by k_bucket_fold (<= 16 chunks) / k_heavy_combine.
  uint32_t L0 = ctx->L0;
  while (L0 < 65536u && ((uint64_t)n * (uint64_t)nlw >> 18) > (uint64_t)L0) L0 <<= 1;
  int rc = ensure(ctx, n, nb_total, nlw, nitems, L0);
  if (rc) return rc;
  hipStream_t st = ctx->stream;
  const uint32_t n32 = (uint32_t)n;
  const uint32_t gn = (n32 + 255) / 256;
  auto h0 = std::chrono::steady_clock::now();
  HIPCHK(hipEventRecord(ctx->ev[0], st));
  hipLaunchKernelGGL(k_prepare_points, dim3(gn), dim3(256), 0, st, (const uint32_t*)d_points96, ctx->d_pts, ctx->d_flags, n32);
  HIPCHK(hipEventRecord(ctx->ev[1], st));
  const uint32_t nblk = (uint32_t)((nb_total + SCAN_ITEMS - 1) / SCAN_ITEMS);
  if (ctx->use_partition_sort && n <= PART_MAX_N) {
    // ---- two-level partition sort: no global atomics
    const uint32_t bb = (uint32_t)c - 1u;                       // bucket bits
    const uint32_t sub_bits = bb < 8u ? bb : 8u, nbins = 1u << (bb - sub_bits);
    const uint32_t nslices = (n32 + PART_TILE - 1) / PART_TILE;
    const uint32_t nbc = (uint32_t)nlw * nbins * nslices;
    hipLaunchKernelGGL(k_digits, dim3(gn), dim3(256), 0, st, (const uint32_t*)d_scalars32, ctx->d_flags, ctx->d_digits, n32, c, nwin, rank, world);
    hipLaunchKernelGGL(k_part_count, dim3(nslices, nlw), dim3(256), 0, st, ctx->d_digits, ctx->d_blockcnt, n32, nslices, nbins, sub_bits);
    const uint32_t ublk = (nbc + SCAN_ITEMS - 1) / SCAN_ITEMS;
    hipLaunchKernelGGL(k_uscan1, dim3(ublk), dim3(256), 0, st, ctx->d_blockcnt, ctx->d_ublocktot, nbc);
    hipLaunchKernelGGL(k_uscan2, dim3(1), dim3(256), 0, st, ctx->d_ublocktot, ublk, ctx->d_blockcnt, nbc);
    hipLaunchKernelGGL(k_uscan3, dim3(ublk), dim3(256), 0, st, ctx->d_ublocktot, ctx->d_blockcnt, nbc);
    HIPCHK(hipEventRecord(ctx->ev[2], st));
    hipLaunchKernelGGL(k_part_scatter, dim3(nslices, nlw), dim3(256), 0, st, ctx->d_digits, ctx->d_blockcnt, ctx->d_part, n32, nslices, nbins, sub_bits);
    hipLaunchKernelGGL(k_bin_sort, dim3((uint32_t)nlw * nbins), dim3(256), 0, st, ctx->d_part, ctx->d_blockcnt, ctx->d_hist, ctx->d_sorted, (uint32_t)nlw * nbins, nslices, sub_bits);
    HIPCHK(hipEventRecord(ctx->ev[3], st));
    hipLaunchKernelGGL(k_scan1, dim3(nblk), dim3(256), 0, st, ctx->d_hist, ctx->d_off, ctx->d_choff, ctx->d_blocktot, (uint32_t)nb_total, L0);
    hipLaunchKernelGGL(k_scan2, dim3(1), dim3(256), 0, st, ctx->d_blocktot, nblk, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
    hipLaunchKernelGGL(k_scan3, dim3(nblk), dim3(256), 0, st, ctx->d_blocktot, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
  } else {
    // ---- global-atomic counting sort (any n < 2^31)
    HIPCHK(hipMemsetAsync(ctx->d_hist, 0, nb_total * 4, st));
    hipLaunchKernelGGL(k_hist, dim3(gn), dim3(256), 0, st, (const uint32_t*)d_scalars32, ctx->d_flags, ctx->d_hist, n32, c, nwin, rank, world);
    HIPCHK(hipEventRecord(ctx->ev[2], st));
    hipLaunchKernelGGL(k_scan1, dim3(nblk), dim3(256), 0, st, ctx->d_hist, ctx->d_off, ctx->d_choff, ctx->d_blocktot, (uint32_t)nb_total, L0);
    hipLaunchKernelGGL(k_scan2, dim3(1), dim3(256), 0, st, ctx->d_blocktot, nblk, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
    hipLaunchKernelGGL(k_scan3, dim3(nblk), dim3(256), 0, st, ctx->d_blocktot, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
    HIPCHK(hipEventRecord(ctx->ev[3], st));
    hipLaunchKernelGGL(k_scatter, dim3(gn), dim3(256), 0, st, (const uint32_t*)d_scalars32, ctx->d_flags, ctx->d_hist, ctx->d_off, ctx->d_sorted, n32, c, nwin, rank, world);
  }
  HIPCHK(hipMemsetAsync(ctx->d_lenhist, 0, LEN_BINS * 4, st));
  HIPCHK(hipMemsetAsync(ctx->d_heavy, 0, 4, st));
  HIPCHK(hipMemsetAsync(ctx->d_combined, 0, nb_total, st));
  hipLaunchKernelGGL(k_chunk_desc, dim3((uint32_t)((nb_total + 255) / 256)), dim3(256), 0, st, ctx->d_off, ctx->d_choff, ctx->d_desc, ctx->d_lenhist, ctx->d_heavy, (uint32_t)ctx->cap_heavy, (uint32_t)nb_total, L0);
  const size_t max_chunks = nb_total + (n * (size_t)nlw) / L0 + 1;
  const uint32_t gchunks = (uint32_t)((max_chunks + 255) / 256);
  hipLaunchKernelGGL(k_len_scan, dim3(1), dim3(256), 0, st, ctx->d_lenhist, ctx->d_lenhist + LEN_BINS);
  hipLaunchKernelGGL(k_order, dim3(gchunks), dim3(256), 0, st, ctx->d_desc, ctx->d_choff + nb_total, ctx->d_lenhist + LEN_BINS, ctx->d_order);
  HIPCHK(hipEventRecord(ctx->ev[4], st));
  hipLaunchKernelGGL(k_accumulate, dim3(gchunks), dim3(256), 0, st, ctx->d_desc, ctx->d_choff + nb_total, ctx->d_order, ctx->d_sorted, ctx->d_pts, ctx->d_sums);
  HIPCHK(hipEventRecord(ctx->ev[5], st));
  hipLaunchKernelGGL(k_heavy_combine, dim3(512), dim3(256), 0, st, ctx->d_heavy, (uint32_t)ctx->cap_heavy, ctx->d_choff, ctx->d_sums, ctx->d_combined);
  hipLaunchKernelGGL(k_bucket_fold, dim3((uint32_t)((nb_total + 255) / 256)), dim3(256), 0, st, ctx->d_choff, ctx->d_sums, ctx->d_combined, (uint32_t)nb_total);
  if (use2d) {
    const uint32_t R = 1u << hb2, Cn = 1u << lb2;
    const uint32_t lpr = Cn < 32u ? Cn : 32u, lpc = R < 16u ? R : 16u;
    const uint32_t nrow_blocks = ((uint32_t)nlw * R + (256u / lpr) - 1u) / (256u / lpr);
    const uint32_t ncol_blocks = ((uint32_t)nlw * Cn + (256u / lpc) - 1u) / (256u / lpc);
    PointSum* rowsum = ctx->d_segrun;                     // reuse the segment buffers (>= nb_total records each)
    PointSum* colsum = ctx->d_segtot;
    hipLaunchKernelGGL(k_rowcol, dim3(nrow_blocks + ncol_blocks), dim3(256), 0, st, ctx->d_choff, ctx->d_sums, ctx->d_combined,
                       rowsum, colsum, (uint32_t)nlw, hb2, lb2, nrow_blocks);
    HIPCHK(hipEventRecord(ctx->ev[6], st));
    hipLaunchKernelGGL(k_small_tree, dim3(nitems, nlw), dim3(256), 0, st, rowsum, colsum, ctx->d_out, hb2, lb2);
  } else {
    const uint32_t nseg_total = (uint32_t)(nb_total / m);
    hipLaunchKernelGGL(k_seg_reduce, dim3((nseg_total + 255) / 256), dim3(256), 0, st, ctx->d_choff, ctx->d_sums, ctx->d_combined, ctx->d_segrun, ctx->d_segtot, nseg_total, m);
    HIPCHK(hipEventRecord(ctx->ev[6], st));
    uint32_t S = (J + BT_ELEMS - 1) / BT_ELEMS; if (S < 1) S = 1; if (S > 64) S = 64;   // J <= 2^15 / seg_m
    hipLaunchKernelGGL(k_bit_tree, dim3(nitems, nlw, S), dim3(256), 0, st, ctx->d_segrun, ctx->d_segtot, ctx->d_partial, J);
    hipLaunchKernelGGL(k_bit_tree_final, dim3((uint32_t)(nitems * nlw)), dim3(64), 0, st, ctx->d_partial, ctx->d_out, S);
  }
  HIPCHK(hipMemcpyAsync(ctx->h_out, ctx->d_out, (size_t)nlw * nitems * sizeof(PointWords), hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(ctx->ev[7], st));
  auto h1 = std::chrono::steady_clock::now();
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  auto h2 = std::chrono::steady_clock::now();
  if (ctx->profile)
    for (int i = 0; i < CG1_NPHASE; ++i) HIPCHK(hipEventElapsedTime(&ctx->phase_ms[i], ctx->ev[i], ctx->ev[i + 1]));
  ctx->last_c = c;

  // ---- host tail: ONE Horner over global bit positions.
  //   result = sum_w 2^(c w) [ T_w + m * sum_b 2^b Y_{w,b} ] = sum over points P with weight 2^e(P),
  //   e(T_w) = c w,  e(Y_{w,b}) = c w + log2(m) + b  (< c (w+1) since log2(m) + nbits = c - 1).
  auto t0 = std::chrono::steady_clock::now();
  ctx->host_ms[0] = std::chrono::duration<float, std::milli>(h1 - h0).count();
  ctx->host_ms[1] = std::chrono::duration<float, std::milli>(h2 - h1).count();
  ctx->host_ms[2] = std::chrono::duration<float, std::milli>(t0 - h2).count();
  int lm = 0; while ((1u << lm) < m) ++lm;
  cg1h::jac acc = cg1h::jac_identity();
  const int top_w = rank + (nlw - 1) * world;
  for (int e = c * top_w + c - 2; e >= 0; --e) {
    acc = cg1h::jac_dbl(acc);
    const int w = e / c, r = e % c;
    if (w % world != rank) continue;
    const PointWords* row = ctx->h_out + (size_t)(w / world) * nitems;
    if (r == 0) acc = cg1h::jac_add(acc, jac_from_words(row[0]));
    if (use2d) {
      //   e(T0) = c w;  e(column bit k) = c w + k (k < lb);  e(row bit k) = c w + lb + k (k < hb)
      if (r < (int)lb2) acc = cg1h::jac_add(acc, jac_from_words(row[1 + hb2 + r]));
      else if (r - (int)lb2 < (int)hb2) acc = cg1h::jac_add(acc, jac_from_words(row[1 + (r - lb2)]));
    } else {
      //   e(T_w) = c w;  e(Y_{w,b}) = c w + log2(m) + b
      if (r >= lm && r - lm < nbits) acc = cg1h::jac_add(acc, jac_from_words(row[1 + (r - lm)]));
    }
  }
  result = acc;
  ctx->host_tail_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx->host_ms[3] = ctx->host_tail_ms;
  return CG1_OK;
}


int pick_window_batched(size_t n_avg) {
  int best = 4; double best_cost = 1e300;
  for (int c = 4; c <= 9; ++c) {                       // NB <= 256: a group's counting sort fits one block's LDS
    if (255 % c == 0) continue;                        // top window would hold only the recoding carry: one hot bucket
    int nwin = 255 / c + 1;
    double NB = (double)(1u << (c - 1));
    double cost = (double)nwin * ((double)n_avg + 1.4 * (2.0 * NB + 3.0 * NB / 8.0)) + 1.4 * 255.0;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

// M independent MSMs over one concatenated (points, scalars) input resident on the device.
int msm_batched_device(Ctx* ctx, const void* d_points96, const void* d_scalars32, const uint32_t* h_offsets, size_t M,
                       int c, std::vector<cg1h::jac>& results) {
  results.assign(M, cg1h::jac_identity());
  if (M == 0) return CG1_OK;
  const size_t N = h_offsets[M];
  for (size_t j = 0; j < M; ++j) if (h_offsets[j] > h_offsets[j + 1]) { snprintf(ctx->err, sizeof ctx->err, "offsets not monotone"); return CG1_ERR_ARG; }
  if (h_offsets[0] != 0) { snprintf(ctx->err, sizeof ctx->err, "offsets[0] must be 0"); return CG1_ERR_ARG; }
  if (N == 0) return CG1_OK;
  if (N >= (1ull << 31) || M > 65535) { snprintf(ctx->err, sizeof ctx->err, "batch too large"); return CG1_ERR_ARG; }
  if (c <= 0) c = pick_window_batched((N + M - 1) / M);
  if (c < 4 || c > 9) { snprintf(ctx->err, sizeof ctx->err, "batched window width %d out of range [4,9]", c); return CG1_ERR_ARG; }
  HIPCHK(hipSetDevice(ctx->device));
  const uint32_t nwin = 255 / c + 1, NB = 1u << (c - 1);
  const size_t G = M * nwin, nb_total = G * NB;
  if (nb_total >= (1ull << 31) || N * nwin >= (1ull << 31)) { snprintf(ctx->err, sizeof ctx->err, "batch too large"); return CG1_ERR_ARG; }
  const uint32_t m = 8 < NB ? 8 : NB;                 // segment length of k_seg_reduce
  uint32_t log2m = 0; while ((1u << log2m) < m) ++log2m;
  const uint32_t J = NB / m;
  uint32_t L0 = ctx->L0;
  while (L0 < 65536u && ((uint64_t)N * (uint64_t)nwin >> 18) > (uint64_t)L0) L0 <<= 1;
  int rc = ensure(ctx, N, nb_total, nwin, 1, L0);
  if (rc) return rc;
  // batch-only buffers
  if ((M + 1) > ctx->cap_boffs) {
    if (ctx->d_boffs) (void)hipFree(ctx->d_boffs);
    HIPCHK(hipMalloc(&ctx->d_boffs, (M + 1) * 4));
    ctx->cap_boffs = M + 1;
  }
  if (G > ctx->cap_gsum) {
    if (ctx->d_gsum) (void)hipFree(ctx->d_gsum);
    HIPCHK(hipMalloc(&ctx->d_gsum, G * sizeof(PointSum)));
    ctx->cap_gsum = G;
  }
  if (M > ctx->cap_bout) {
    if (ctx->d_bout) (void)hipFree(ctx->d_bout);
    if (ctx->h_bout) (void)hipHostFree(ctx->h_bout);
    HIPCHK(hipMalloc(&ctx->d_bout, M * sizeof(PointWords)));
    HIPCHK(hipHostMalloc(&ctx->h_bout, M * sizeof(PointWords)));
    ctx->cap_bout = M;
  }
  if (N * nwin > ctx->cap_digits) {
    if (ctx->d_digits) (void)hipFree(ctx->d_digits);
    HIPCHK(hipMalloc(&ctx->d_digits, N * nwin * 2 + 16));
    ctx->cap_digits = N * nwin;
  }
  hipStream_t st = ctx->stream;
  const uint32_t N32 = (uint32_t)N, gn = (N32 + 255) / 256;
  auto h0 = std::chrono::steady_clock::now();
  HIPCHK(hipMemcpyAsync(ctx->d_boffs, h_offsets, (M + 1) * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(ctx->ev[0], st));
  hipLaunchKernelGGL(k_prepare_points, dim3(gn), dim3(256), 0, st, (const uint32_t*)d_points96, ctx->d_pts, ctx->d_flags, N32);
  HIPCHK(hipEventRecord(ctx->ev[1], st));
  hipLaunchKernelGGL(k_digits, dim3(gn), dim3(256), 0, st, (const uint32_t*)d_scalars32, ctx->d_flags, ctx->d_digits, N32, c, (int)nwin, 0, 1);
  hipLaunchKernelGGL(k_group_count, dim3((uint32_t)M, nwin), dim3(256), 0, st, ctx->d_digits, ctx->d_boffs, ctx->d_hist, N32, NB, nwin);
  HIPCHK(hipEventRecord(ctx->ev[2], st));
  const uint32_t nblk = (uint32_t)((nb_total + SCAN_ITEMS - 1) / SCAN_ITEMS);
  hipLaunchKernelGGL(k_scan1, dim3(nblk), dim3(256), 0, st, ctx->d_hist, ctx->d_off, ctx->d_choff, ctx->d_blocktot, (uint32_t)nb_total, L0);
  hipLaunchKernelGGL(k_scan2, dim3(1), dim3(256), 0, st, ctx->d_blocktot, nblk, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
  hipLaunchKernelGGL(k_scan3, dim3(nblk), dim3(256), 0, st, ctx->d_blocktot, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
  hipLaunchKernelGGL(k_group_scatter, dim3((uint32_t)M, nwin), dim3(256), 0, st, ctx->d_digits, ctx->d_boffs, ctx->d_off, ctx->d_sorted, N32, NB, nwin);
  HIPCHK(hipEventRecord(ctx->ev[3], st));
  HIPCHK(hipMemsetAsync(ctx->d_lenhist, 0, LEN_BINS * 4, st));
  HIPCHK(hipMemsetAsync(ctx->d_heavy, 0, 4, st));
  HIPCHK(hipMemsetAsync(ctx->d_combined, 0, nb_total, st));
  hipLaunchKernelGGL(k_chunk_desc, dim3((uint32_t)((nb_total + 255) / 256)), dim3(256), 0, st, ctx->d_off, ctx->d_choff, ctx->d_desc, ctx->d_lenhist, ctx->d_heavy, (uint32_t)ctx->cap_heavy, (uint32_t)nb_total, L0);
  const size_t max_chunks = nb_total + (N * (size_t)nwin) / L0 + 1;
  const uint32_t gchunks = (uint32_t)((max_chunks + 255) / 256);
  hipLaunchKernelGGL(k_len_scan, dim3(1), dim3(256), 0, st, ctx->d_lenhist, ctx->d_lenhist + LEN_BINS);
  hipLaunchKernelGGL(k_order, dim3(gchunks), dim3(256), 0, st, ctx->d_desc, ctx->d_choff + nb_total, ctx->d_lenhist + LEN_BINS, ctx->d_order);
  HIPCHK(hipEventRecord(ctx->ev[4], st));
  hipLaunchKernelGGL(k_accumulate, dim3(gchunks), dim3(256), 0, st, ctx->d_desc, ctx->d_choff + nb_total, ctx->d_order, ctx->d_sorted, ctx->d_pts, ctx->d_sums);
  HIPCHK(hipEventRecord(ctx->ev[5], st));
  const uint32_t nseg_total = (uint32_t)(nb_total / m);
  hipLaunchKernelGGL(k_heavy_combine, dim3(512), dim3(256), 0, st, ctx->d_heavy, (uint32_t)ctx->cap_heavy, ctx->d_choff, ctx->d_sums, ctx->d_combined);
  hipLaunchKernelGGL(k_seg_reduce, dim3((nseg_total + 255) / 256), dim3(256), 0, st, ctx->d_choff, ctx->d_sums, ctx->d_combined, ctx->d_segrun, ctx->d_segtot, nseg_total, m);
  hipLaunchKernelGGL(k_group_reduce, dim3((uint32_t)((G + 255) / 256)), dim3(256), 0, st, ctx->d_segrun, ctx->d_segtot, ctx->d_gsum, (uint32_t)G, J, log2m);
  HIPCHK(hipEventRecord(ctx->ev[6], st));
  hipLaunchKernelGGL(k_msm_horner, dim3((uint32_t)((M + 63) / 64)), dim3(64), 0, st, ctx->d_gsum, ctx->d_bout, (uint32_t)M, nwin, (uint32_t)c);
  HIPCHK(hipMemcpyAsync(ctx->h_bout, ctx->d_bout, M * sizeof(PointWords), hipMemcpyDeviceToHost, st));
  HIPCHK(hipEventRecord(ctx->ev[7], st));
  auto h1 = std::chrono::steady_clock::now();
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  auto h2 = std::chrono::steady_clock::now();
  if (ctx->profile)
    for (int i = 0; i < CG1_NPHASE; ++i) HIPCHK(hipEventElapsedTime(&ctx->phase_ms[i], ctx->ev[i], ctx->ev[i + 1]));
  ctx->last_c = c;
  for (size_t j = 0; j < M; ++j) results[j] = jac_from_words(ctx->h_bout[j]);
  auto h3 = std::chrono::steady_clock::now();
  ctx->host_ms[0] = std::chrono::duration<float, std::milli>(h1 - h0).count();
  ctx->host_ms[1] = std::chrono::duration<float, std::milli>(h2 - h1).count();
  ctx->host_ms[2] = 0;
  ctx->host_ms[3] = ctx->host_tail_ms = std::chrono::duration<float, std::milli>(h3 - h2).count();
  return CG1_OK;
}

}  // namespace cg1

// ================================================================== C ABI (include/curdle_g1.h)
using cg1::Ctx;
struct cg1_ctx : public cg1::Ctx {};

static inline cg1h::jac blob_in(const uint8_t* b) { cg1h::jac j; memcpy(&j, b, sizeof j); return j; }
static inline void blob_out(uint8_t* b, const cg1h::jac& j) { memcpy(b, &j, sizeof j); }
static_assert(sizeof(cg1h::jac) == CG1_POINT_BYTES, "point blob size");

extern "C" {

void cg1_identity(uint8_t* out) { blob_out(out, cg1h::jac_identity()); }
void cg1_generator(uint8_t* out) { blob_out(out, cg1h::jac_generator()); }
void cg1_add(uint8_t* out, const uint8_t* a, const uint8_t* b) { blob_out(out, cg1h::jac_add(blob_in(a), blob_in(b))); }
void cg1_sub(uint8_t* out, const uint8_t* a, const uint8_t* b) { blob_out(out, cg1h::jac_add(blob_in(a), cg1h::jac_neg(blob_in(b)))); }
void cg1_neg(uint8_t* out, const uint8_t* a) { blob_out(out, cg1h::jac_neg(blob_in(a))); }
void cg1_double(uint8_t* out, const uint8_t* a) { blob_out(out, cg1h::jac_dbl(blob_in(a))); }
void cg1_mul(uint8_t* out, const uint8_t* a, const uint8_t* k) { blob_out(out, cg1h::jac_mul(blob_in(a), k)); }
int cg1_eq(const uint8_t* a, const uint8_t* b) { return cg1h::jac_eq(blob_in(a), blob_in(b)) ? 1 : 0; }
int cg1_is_identity(const uint8_t* a) { return cg1h::jac_is_identity(blob_in(a)) ? 1 : 0; }
void cg1_compress(uint8_t* out48, const uint8_t* a) { cg1h::g1_compress(blob_in(a), out48); }
static int map_dec(int rc) {
  switch (rc) { case 0: return CG1_OK; case 1: case 2: return CG1_ERR_ENCODING; case 3: return CG1_ERR_NOT_ON_CURVE; default: return CG1_ERR_NOT_IN_SUBGROUP; }
}
int cg1_decompress(uint8_t* out, const uint8_t* in48, int check_subgroup) {
  cg1h::jac j;
  int rc = cg1h::g1_decompress(in48, check_subgroup != 0, j);
  if (rc == 0) blob_out(out, j);
  return map_dec(rc);
}
void cg1_to_affine96(uint8_t* out96, const uint8_t* a) {
  cg1h::fe x, y; bool inf;
  cg1h::jac_to_affine(blob_in(a), x, y, inf);
  if (inf) { memset(out96, 0, 96); return; }
  cg1h::fe_to_le48(x, out96); cg1h::fe_to_le48(y, out96 + 48);
}
int cg1_from_affine96(uint8_t* out, const uint8_t* in96, int check_on_curve) {
  bool any = false;
  for (int i = 0; i < 96; ++i) any = any || in96[i];
  if (!any) { blob_out(out, cg1h::jac_identity()); return CG1_OK; }
  cg1h::fe x, y;
  if (!cg1h::fe_from_le48(in96, x) || !cg1h::fe_from_le48(in96 + 48, y)) return CG1_ERR_ENCODING;
  cg1h::jac j = cg1h::jac_from_affine(x, y);
  if (check_on_curve && !cg1h::jac_on_curve(j)) return CG1_ERR_NOT_ON_CURVE;
  blob_out(out, j);
  return CG1_OK;
}
void cg1_batch_to_affine96(uint8_t* out96, const uint8_t* blobs, size_t n) {
  std::vector<cg1h::jac> pts(n);
  std::vector<cg1h::fe> xs(n), ys(n);
  std::vector<uint8_t> inf(n);
  for (size_t i = 0; i < n; ++i) pts[i] = blob_in(blobs + CG1_POINT_BYTES * i);
  cg1h::jac_batch_to_affine(pts.data(), n, xs.data(), ys.data(), inf.data());
  for (size_t i = 0; i < n; ++i) {
    uint8_t* o = out96 + 96 * i;
    if (inf[i]) { memset(o, 0, 96); continue; }
    cg1h::fe_to_le48(xs[i], o); cg1h::fe_to_le48(ys[i], o + 48);
  }
}
int cg1_batch_decompress(uint8_t* out_blobs, const uint8_t* in48, size_t n, int check_subgroup, size_t* bad_index) {
  for (size_t i = 0; i < n; ++i) {
    int rc = cg1_decompress(out_blobs + CG1_POINT_BYTES * i, in48 + 48 * i, check_subgroup);
    if (rc) { if (bad_index) *bad_index = i; return rc; }
  }
  return CG1_OK;
}
void cg1_batch_compress(uint8_t* out48, const uint8_t* blobs, size_t n) {
  std::vector<cg1h::jac> pts(n);
  std::vector<cg1h::fe> xs(n), ys(n);
  std::vector<uint8_t> inf(n);
  for (size_t i = 0; i < n; ++i) pts[i] = blob_in(blobs + CG1_POINT_BYTES * i);
  cg1h::jac_batch_to_affine(pts.data(), n, xs.data(), ys.data(), inf.data());
  for (size_t i = 0; i < n; ++i) {
    uint8_t* o = out48 + 48 * i;
    if (inf[i]) { memset(o, 0, 48); o[0] = 0xC0; continue; }
    cg1h::fe_to_be48(xs[i], o);
    o[0] |= 0x80;
    if (cg1h::fe_lex_largest(ys[i])) o[0] |= 0x20;
  }
}

// ---------------------------------------------------------------- device
int cg1_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

cg1_ctx* cg1_ctx_create(int device) {
  int n = cg1_device_count();
  if (device < 0 || device >= n) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  cg1_ctx* ctx = new cg1_ctx();
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return nullptr; }
  for (int i = 0; i <= CG1_NPHASE; ++i) if (hipEventCreate(&ctx->ev[i]) != hipSuccess) { delete ctx; return nullptr; }
  return ctx;
}
void cg1_ctx_destroy(cg1_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  cg1::free_bufs(ctx);
  if (ctx->d_stage_pts) (void)hipFree(ctx->d_stage_pts);
  if (ctx->d_stage_sc) (void)hipFree(ctx->d_stage_sc);
  for (int i = 0; i <= CG1_NPHASE; ++i) (void)hipEventDestroy(ctx->ev[i]);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}
const char* cg1_ctx_error(const cg1_ctx* ctx) { return ctx ? ctx->err : "null context (no GPU visible?)"; }

void* cg1_dev_malloc(cg1_ctx* ctx, size_t bytes) {
  if (!ctx) return nullptr;
  void* p = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
  return p;
}
void cg1_dev_free(cg1_ctx* ctx, void* p) { if (ctx && p) { (void)hipSetDevice(ctx->device); (void)hipFree(p); } }
int cg1_h2d(cg1_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return CG1_ERR_HIP;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return CG1_OK;
}
int cg1_d2h(cg1_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return CG1_ERR_HIP;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return CG1_OK;
}
int cg1_ctx_sync(cg1_ctx* ctx) {
  if (!ctx) return CG1_ERR_HIP;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());
  return CG1_OK;
}
int cg1_ctx_set_param(cg1_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return CG1_ERR_ARG;
  if (!strcmp(name, "chunk_len")) { if (value < 1 || value > 65536) return CG1_ERR_ARG; ctx->L0 = (uint32_t)value; cg1::free_bufs(ctx); return CG1_OK; }
  if (!strcmp(name, "reduce_2d")) { ctx->reduce_2d = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "partition_sort")) { ctx->use_partition_sort = value ? 1 : 0; cg1::free_bufs(ctx); return CG1_OK; }
  if (!strcmp(name, "wave_agg")) {
    int v = value ? 1 : 0;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(cg1::g_wave_agg), &v, sizeof v));
    return CG1_OK;
  }
  if (!strcmp(name, "profile")) { ctx->profile = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "seg_m")) { if (value != 1 && value != 2 && value != 4 && value != 8 && value != 16) return CG1_ERR_ARG; ctx->seg_m = (uint32_t)value; return CG1_OK; }
  return CG1_ERR_ARG;
}

int cg1_msm_device(cg1_ctx* ctx, const void* d_points, const void* d_scalars, size_t n, int window_c, int shard_rank,
                   int shard_world, uint8_t* out) {
  if (!ctx) return CG1_ERR_HIP;
  cg1h::jac r;
  int rc = cg1::msm_device(ctx, d_points, d_scalars, n, window_c, shard_rank, shard_world, r);
  if (rc == CG1_OK) blob_out(out, r);
  return rc;
}

int cg1_msm(cg1_ctx* ctx, const uint8_t* points, const uint8_t* scalars, size_t n, uint8_t* out) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) { blob_out(out, cg1h::jac_identity()); return CG1_OK; }
  HIPCHK(hipSetDevice(ctx->device));
  if (n > ctx->cap_stage) {
    if (ctx->d_stage_pts) (void)hipFree(ctx->d_stage_pts);
    if (ctx->d_stage_sc) (void)hipFree(ctx->d_stage_sc);
    ctx->d_stage_pts = ctx->d_stage_sc = nullptr; ctx->cap_stage = 0;
    HIPCHK(hipMalloc(&ctx->d_stage_pts, n * 96));
    HIPCHK(hipMalloc(&ctx->d_stage_sc, n * 32));
    ctx->cap_stage = n;
  }
  HIPCHK(hipMemcpyAsync(ctx->d_stage_pts, points, n * 96, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_stage_sc, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
  return cg1_msm_device(ctx, ctx->d_stage_pts, ctx->d_stage_sc, n, 0, 0, 1, out);
}

int cg1_msm_batched_device(cg1_ctx* ctx, const void* d_points, const void* d_scalars, const uint32_t* offsets, size_t n_msm,
                           int window_c, uint8_t* out_blobs) {
  if (!ctx) return CG1_ERR_HIP;
  if (!offsets && n_msm) return CG1_ERR_ARG;
  std::vector<cg1h::jac> res;
  int rc = cg1::msm_batched_device(ctx, d_points, d_scalars, offsets, n_msm, window_c, res);
  if (rc == CG1_OK) for (size_t j = 0; j < n_msm; ++j) blob_out(out_blobs + CG1_POINT_BYTES * j, res[j]);
  return rc;
}

int cg1_msm_batched(cg1_ctx* ctx, const uint8_t* points, const uint8_t* scalars, const uint32_t* offsets, size_t n_msm,
                    uint8_t* out_blobs) {
  if (!ctx) return CG1_ERR_HIP;
  if (n_msm == 0) return CG1_OK;
  if (!offsets) return CG1_ERR_ARG;
  const size_t n = offsets[n_msm];
  if (n == 0) { for (size_t j = 0; j < n_msm; ++j) blob_out(out_blobs + CG1_POINT_BYTES * j, cg1h::jac_identity()); return CG1_OK; }
  HIPCHK(hipSetDevice(ctx->device));
  if (n > ctx->cap_stage) {
    if (ctx->d_stage_pts) (void)hipFree(ctx->d_stage_pts);
    if (ctx->d_stage_sc) (void)hipFree(ctx->d_stage_sc);
    ctx->d_stage_pts = ctx->d_stage_sc = nullptr; ctx->cap_stage = 0;
    HIPCHK(hipMalloc(&ctx->d_stage_pts, n * 96));
    HIPCHK(hipMalloc(&ctx->d_stage_sc, n * 32));
    ctx->cap_stage = n;
  }
  HIPCHK(hipMemcpyAsync(ctx->d_stage_pts, points, n * 96, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_stage_sc, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
  return cg1_msm_batched_device(ctx, ctx->d_stage_pts, ctx->d_stage_sc, offsets, n_msm, 0, out_blobs);
}

int cg1_get_timings(const cg1_ctx* ctx, float* phase_ms, float* host_tail_ms, int* window_c) {
  if (!ctx) return CG1_ERR_ARG;
  if (phase_ms) for (int i = 0; i < CG1_NPHASE; ++i) phase_ms[i] = ctx->phase_ms[i];
  if (host_tail_ms) *host_tail_ms = ctx->host_tail_ms;
  if (window_c) *window_c = ctx->last_c;
  return CG1_OK;
}

int cg1_get_host_timings(const cg1_ctx* ctx, float host_ms[4]) {
  if (!ctx || !host_ms) return CG1_ERR_ARG;
  for (int i = 0; i < 4; ++i) host_ms[i] = ctx->host_ms[i];
  return CG1_OK;
}

int cg1_batch_mul_add_device(cg1_ctx* ctx, const void* d_bases, size_t nbase, const void* d_scalars, size_t nscalars,
                             const void* d_addend, void* d_out, size_t n) {
  if (!ctx) return CG1_ERR_HIP;
  if ((nbase == 0 || nscalars == 0) && n) return CG1_ERR_ARG;
  if (n == 0) return CG1_OK;
  if (n >= (1ull << 31)) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(cg1::k_batch_mul, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream,
                     (const uint32_t*)d_bases, (uint32_t)nbase, (const uint32_t*)d_scalars, (uint32_t)nscalars,
                     (const uint32_t*)d_addend, (uint32_t*)d_out, (uint32_t)n);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  return CG1_OK;
}
int cg1_batch_mul_device(cg1_ctx* ctx, const void* d_bases, size_t nbase, const void* d_scalars, void* d_out, size_t n) {
  return cg1_batch_mul_add_device(ctx, d_bases, nbase, d_scalars, n ? n : 1, nullptr, d_out, n);
}
// host-pointer convenience: H2D, kernel, D2H
int cg1_batch_mul_add(cg1_ctx* ctx, const uint8_t* bases, size_t nbase, const uint8_t* scalars, size_t nscalars,
                      const uint8_t* addend, uint8_t* out, size_t n) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) return CG1_OK;
  if (nbase == 0 || nscalars == 0) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  void *db = nullptr, *ds = nullptr, *da = nullptr, *dout = nullptr;
  HIPCHK(hipMalloc(&db, nbase * 96)); HIPCHK(hipMalloc(&ds, nscalars * 32)); HIPCHK(hipMalloc(&dout, n * 96));
  HIPCHK(hipMemcpy(db, bases, nbase * 96, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ds, scalars, nscalars * 32, hipMemcpyHostToDevice));
  if (addend) { HIPCHK(hipMalloc(&da, n * 96)); HIPCHK(hipMemcpy(da, addend, n * 96, hipMemcpyHostToDevice)); }
  int rc = cg1_batch_mul_add_device(ctx, db, nbase, ds, nscalars, da, dout, n);
  if (rc == CG1_OK) { hipError_t e = hipMemcpy(out, dout, n * 96, hipMemcpyDeviceToHost); if (e != hipSuccess) rc = CG1_ERR_HIP; }
  (void)hipFree(db); (void)hipFree(ds); (void)hipFree(dout); if (da) (void)hipFree(da);
  return rc;
}
// n compressed48 (device) -> n affine96 + n status bytes (device); returns CG1_OK when the kernel ran
int cg1_batch_decompress_device(cg1_ctx* ctx, const void* d_in48, void* d_out_affine96, void* d_status, size_t n, int check_subgroup) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) return CG1_OK;
  if (n >= (1ull << 31)) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(cg1::k_batch_decompress, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream,
                     (const uint8_t*)d_in48, (uint32_t*)d_out_affine96, (uint8_t*)d_status, (uint32_t)n, check_subgroup);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  return CG1_OK;
}
// host buffers: returns CG1_OK if every encoding is valid, else the first failing status with *bad_index set
int cg1_batch_decompress_gpu(cg1_ctx* ctx, const uint8_t* in48, uint8_t* out_affine96, size_t n, int check_subgroup, size_t* bad_index) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) return CG1_OK;
  HIPCHK(hipSetDevice(ctx->device));
  void *din = nullptr, *dout = nullptr, *dst = nullptr;
  HIPCHK(hipMalloc(&din, n * 48)); HIPCHK(hipMalloc(&dout, n * 96)); HIPCHK(hipMalloc(&dst, n));
  HIPCHK(hipMemcpy(din, in48, n * 48, hipMemcpyHostToDevice));
  int rc = cg1_batch_decompress_device(ctx, din, dout, dst, n, check_subgroup);
  std::vector<uint8_t> st(n);
  if (rc == CG1_OK) {
    if (hipMemcpy(out_affine96, dout, n * 96, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(st.data(), dst, n, hipMemcpyDeviceToHost) != hipSuccess) rc = CG1_ERR_HIP;
  }
  (void)hipFree(din); (void)hipFree(dout); (void)hipFree(dst);
  if (rc != CG1_OK) return rc;
  for (size_t i = 0; i < n; ++i) if (st[i]) { if (bad_index) *bad_index = i; return st[i]; }
  return CG1_OK;
}

int cg1_gen_scalars_device(cg1_ctx* ctx, void* d_out, size_t n, uint64_t seed) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) return CG1_OK;
  HIPCHK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(cg1::k_gen_scalars, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (uint32_t*)d_out, (uint32_t)n, seed);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  return CG1_OK;
}
int cg1_probe_madd(cg1_ctx* ctx, const void* d_points, size_t npts, size_t lanes, int iters, float* ms) {
  if (!ctx) return CG1_ERR_HIP;
  if (npts == 0 || lanes == 0 || lanes % 256) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  cg1::PreparedPoint* prep = nullptr; cg1::PointSum* out = nullptr; uint8_t* fl = nullptr;
  HIPCHK(hipMalloc(&prep, npts * sizeof(cg1::PreparedPoint)));
  HIPCHK(hipMalloc(&fl, npts + 16));
  HIPCHK(hipMalloc(&out, lanes * sizeof(cg1::PointSum)));
  hipLaunchKernelGGL(cg1::k_prepare_points, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_points, prep, fl, (uint32_t)npts);
  hipLaunchKernelGGL(cg1::k_probe_madd, dim3((unsigned)(lanes / 256)), dim3(256), 0, ctx->stream, prep, (uint32_t)npts, out, 2);
  HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
  hipLaunchKernelGGL(cg1::k_probe_madd, dim3((unsigned)(lanes / 256)), dim3(256), 0, ctx->stream, prep, (uint32_t)npts, out, iters);
  HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventElapsedTime(ms, ctx->ev[0], ctx->ev[1]));
  (void)hipFree(prep); (void)hipFree(out); (void)hipFree(fl);
  return CG1_OK;
}

}  // extern "C"
