// Two-level LDS partition sort, regime-B group sort, scans, chunking and chunk-length ordering.
// Part of the single translation unit csrc/msm_gpu.hip (included inside namespace cg1).
#pragma once

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
                                                uint16_t* __restrict__ digits, uint32_t n, WinPlan pl, int rank, int world,
                                                uint32_t* __restrict__ bad_flag) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const bool inf = inf_flag[i] != 0;
  DigitIter it;
  load_scalar(scalars, i, it);
  if (it.s[7] >> 31) *bad_flag = 1u;              // >= 2^255: the signed-digit recoding would carry out of the top window
  if (pl.glv) {
    // two digit rows per scalar: the halves k1 (on P_i, row entry i) and k2 (on phi(P_i), entry n + i) of k = k1 + k2 lambda; both
    // magnitudes are <= (lambda + 1) / 2 < 0.68 * 2^127, so the recoding carry never leaves the top window of a 128-position plan
    GlvParts g;
    glv_split(it.s, g);
    DigitIter a, b;
    load_half(g.k1, g.neg1, a);
    load_half(g.k2, g.neg2, b);
    const size_t nv = 2 * (size_t)n;
    WinWalk ww(rank, world);
    for (int w = 0; w < pl.nwin; ++w) {
      const int d1 = a.next(pl, w), d2 = b.next(pl, w);
      const int lw = ww.step(w);
      if (lw < 0) continue;
      digits[(size_t)lw * nv + i] = inf ? (uint16_t)0 : (uint16_t)(d1 & 0xFFFF);
      digits[(size_t)lw * nv + n + i] = inf ? (uint16_t)0 : (uint16_t)(d2 & 0xFFFF);
    }
    return;
  }
  WinWalk ww(rank, world);
  for (int w = 0; w < pl.nwin; ++w) {
    int d = it.next(pl, w);
    const int lw = ww.step(w);
    if (lw < 0) continue;
    digits[(size_t)lw * n + i] = inf ? (uint16_t)0 : (uint16_t)(d & 0xFFFF);
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
                                                      uint32_t sub_bits, int use_stage) {
  __shared__ uint32_t cur[128];          // staged: local slot cursors; direct: global cursors
  __shared__ uint32_t gbase[128];        // global base of this block's run in each bin
  __shared__ uint32_t lbase[128];        // local exclusive offsets of the bins inside the tile
  __shared__ uint32_t stage[PART_TILE];
  __shared__ uint32_t gpos[PART_TILE];
  const uint32_t slice = blockIdx.x, lw = blockIdx.y;
  uint32_t mycnt = 0;
  if (threadIdx.x < nbins) {
    const size_t idx = ((size_t)lw * nbins + threadIdx.x) * nslices + slice;
    gbase[threadIdx.x] = block_base[idx];
    mycnt = block_base[idx + 1] - block_base[idx];      // exclusive scan in (window, bin, tile) order: next - this = count
    lbase[threadIdx.x] = mycnt;
  } else if (threadIdx.x < 128) {
    lbase[threadIdx.x] = 0;
  }
  __syncthreads();
  if (use_stage) {                                      // exclusive scan of the <= 128 per-bin counts
    for (int d = 1; d < 128; d <<= 1) {
      uint32_t u = (threadIdx.x < 128 && (int)threadIdx.x >= d) ? lbase[threadIdx.x - d] : 0u;
      __syncthreads();
      if (threadIdx.x < 128) lbase[threadIdx.x] += u;
      __syncthreads();
    }
    if (threadIdx.x < 128) { lbase[threadIdx.x] -= mycnt; cur[threadIdx.x] = lbase[threadIdx.x]; }
  } else if (threadIdx.x < nbins) {
    cur[threadIdx.x] = gbase[threadIdx.x];
  }
  __syncthreads();
  const uint16_t* dg = digits + (size_t)lw * n;
  const uint32_t sub_mask = (1u << sub_bits) - 1u;
  for (uint32_t k = 0; k < PART_TILE / 256; ++k) {
    uint32_t i = slice * PART_TILE + k * 256 + threadIdx.x;
    uint32_t e = (i < n) ? (uint32_t)dg[i] : 0u;
    uint32_t neg = 0, b = e ? (digit_mag(e, neg) - 1u) : 0u;
    const uint32_t bin = b >> sub_bits;
    uint32_t pos = lds_ranked_inc(cur, bin, e != 0u);
    const uint32_t val = i | (neg << 23) | ((b & sub_mask) << 24);
    if (e) {
      if (use_stage) { stage[pos] = val; gpos[pos] = gbase[bin] + (pos - lbase[bin]); }
      else part[pos] = val;
    }
  }
  if (use_stage) {
    __syncthreads();
    const uint32_t total = lbase[nbins - 1] + (block_base[((size_t)lw * nbins + nbins - 1) * nslices + slice + 1] - gbase[nbins - 1]);
    for (uint32_t o = threadIdx.x; o < total; o += 256) part[gpos[o]] = stage[o];   // consecutive o of one bin -> consecutive addresses
  }
}
constexpr uint32_t BIN_STAGE = 12288;     // entries of a bin staged in LDS (48 KB) so sorted[] is written as full lines
__global__ void __launch_bounds__(256) k_bin_sort(const uint32_t* __restrict__ part, const uint32_t* __restrict__ block_base,
                                                  uint32_t* __restrict__ hist, uint32_t* __restrict__ sorted,
                                                  uint32_t nbins_total, uint32_t nslices, uint32_t sub_bits, int use_stage,
                                                  const uint8_t* __restrict__ bigflag) {
  __shared__ uint32_t cnt[256];
  __shared__ uint32_t cur[256];
  __shared__ uint32_t stage[BIN_STAGE];
  const uint32_t g = blockIdx.x;
  if (bigflag[g]) return;                                          // oversize bin: sorted slice-wise (k_slice_*)
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
  const bool staged = use_stage && (end - start) <= BIN_STAGE;      // block-uniform
  cur[threadIdx.x] = (staged ? 0u : start) + excl;
  if (threadIdx.x < (1u << sub_bits)) hist[((size_t)g << sub_bits) + threadIdx.x] = mine;
  __syncthreads();
  for (uint32_t o = threadIdx.x; o < span; o += 256) {
    const bool live = start + o < end;
    const uint32_t v = live ? part[start + o] : 0u;
    const uint32_t pos = lds_ranked_inc(cur, v >> 24, live);
    const uint32_t val = (v & 0x7fffffu) | ((v >> 23 & 1u) << 31);
    if (live) { if (staged) stage[pos] = val; else sorted[pos] = val; }
  }
  if (staged) {                                                      // scattered inside LDS, streamed out in order
    __syncthreads();
    for (uint32_t o = threadIdx.x; start + o < end; o += 256) sorted[start + o] = stage[o];
  }
}

// ---- bins too large for k_bin_sort's LDS staging (> BIN_STAGE entries): n > 2^21 (uniform bins of 32 K - 64 K) and
// skewed scalars (all-equal, tiny ranges, recoding-carry windows: up to ALL n entries of a window in one bin, which one
// block would take milliseconds to sort).  Each such bin is cut into slices of <= BIN_STAGE entries, one block per slice:
//   k_slice_plan     slices per bin + their exclusive scan (block -> (bin, slice) map), flags for k_bin_sort to skip
//   k_slice_count    per-slice LDS histogram over the sub-bucket
//   k_slice_prefix   one block per oversize bin: running sums across its slices (in place), per-bucket totals + bases
//   k_slice_scatter  per slice: rank in LDS, stage, write whole runs -> sorted[]
// Fixed upper-bound grids (blocks beyond the planned count exit at once); no host round trip.
constexpr uint32_t SLICE = BIN_STAGE;

__global__ void __launch_bounds__(256) k_slice_plan(const uint32_t* __restrict__ block_base, uint32_t nbins_total, uint32_t nslices,
                                                    uint32_t* __restrict__ slice_base /* [nbins_total + 1] */,
                                                    uint8_t* __restrict__ bigflag, int enable) {
  __shared__ uint32_t part_sum[256];
  // thread t owns bins [t*per, (t+1)*per): serial count, block scan of the 256 partial sums, serial write-back
  const uint32_t per = (nbins_total + 255u) / 256u, g0 = threadIdx.x * per;
  uint32_t mine = 0;
  for (uint32_t g = g0; g < g0 + per && g < nbins_total; ++g) {
    const uint32_t size = block_base[(size_t)(g + 1) * nslices] - block_base[(size_t)g * nslices];
    mine += (enable && size > BIN_STAGE) ? (size + SLICE - 1u) / SLICE : 0u;
  }
  part_sum[threadIdx.x] = mine;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    uint32_t u = ((int)threadIdx.x >= d) ? part_sum[threadIdx.x - d] : 0u;
    __syncthreads();
    part_sum[threadIdx.x] += u;
    __syncthreads();
  }
  uint32_t run = part_sum[threadIdx.x] - mine;
  for (uint32_t g = g0; g < g0 + per && g < nbins_total; ++g) {
    const uint32_t size = block_base[(size_t)(g + 1) * nslices] - block_base[(size_t)g * nslices];
    const uint32_t ns = (enable && size > BIN_STAGE) ? (size + SLICE - 1u) / SLICE : 0u;
    slice_base[g] = run;
    bigflag[g] = ns ? 1 : 0;
    run += ns;
  }
  if (threadIdx.x == 255) slice_base[nbins_total] = part_sum[255];
}

// block b -> (bin g, slice s): the last g with slice_base[g] <= b (bins without slices share their successor's base)
__device__ __forceinline__ bool slice_of_block(const uint32_t* __restrict__ slice_base, uint32_t nbins_total, uint32_t b,
                                               uint32_t& g, uint32_t& s) {
  if (b >= slice_base[nbins_total]) return false;
  uint32_t lo = 0, hi = nbins_total;                    // invariant: slice_base[lo] <= b < slice_base[hi]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (slice_base[mid] <= b) lo = mid; else hi = mid;
  }
  g = lo;
  s = b - slice_base[lo];
  return true;
}

__global__ void __launch_bounds__(256) k_slice_count(const uint32_t* __restrict__ part, const uint32_t* __restrict__ block_base,
                                                     uint32_t nbins_total, uint32_t nslices, const uint32_t* __restrict__ slice_base,
                                                     uint32_t* __restrict__ slicehist) {
  __shared__ uint32_t cnt[256];
  uint32_t g, s;
  if (!slice_of_block(slice_base, nbins_total, blockIdx.x, g, s)) return;
  const uint32_t start = block_base[(size_t)g * nslices], end = block_base[(size_t)(g + 1) * nslices];
  const uint32_t lo = start + s * SLICE, hi = (lo + SLICE < end) ? lo + SLICE : end;
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t span = ((hi - lo + 255u) / 256u) * 256u;          // whole waves iterate together
  for (uint32_t o = threadIdx.x; o < span; o += 256) {
    const bool live = lo + o < hi;
    lds_ranked_inc(cnt, live ? (part[lo + o] >> 24) : 0u, live);
  }
  __syncthreads();
  slicehist[(size_t)blockIdx.x * 256u + threadIdx.x] = cnt[threadIdx.x];
}

// one block per bin; thread t = sub-bucket t.  slicehist[slice][t] becomes the number of entries of bucket t in the
// EARLIER slices of the bin; subbase[g][t] = where bucket t of bin g starts in sorted[]; hist[] gets the bucket totals.
__global__ void __launch_bounds__(256) k_slice_prefix(const uint32_t* __restrict__ block_base, uint32_t nslices, uint32_t sub_bits,
                                                      const uint32_t* __restrict__ slice_base, const uint8_t* __restrict__ bigflag,
                                                      uint32_t* __restrict__ slicehist, uint32_t* __restrict__ subbase,
                                                      uint32_t* __restrict__ hist) {
  __shared__ uint32_t tot[256];
  const uint32_t g = blockIdx.x;
  if (!bigflag[g]) return;
  const uint32_t b0 = slice_base[g], b1 = slice_base[g + 1];
  uint32_t run = 0;
  for (uint32_t b = b0; b < b1; ++b) {
    const size_t idx = (size_t)b * 256u + threadIdx.x;
    const uint32_t v = slicehist[idx];
    slicehist[idx] = run;
    run += v;
  }
  tot[threadIdx.x] = run;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    uint32_t u = ((int)threadIdx.x >= d) ? tot[threadIdx.x - d] : 0u;
    __syncthreads();
    tot[threadIdx.x] += u;
    __syncthreads();
  }
  subbase[(size_t)g * 256u + threadIdx.x] = block_base[(size_t)g * nslices] + tot[threadIdx.x] - run;
  if (threadIdx.x < (1u << sub_bits)) hist[((size_t)g << sub_bits) + threadIdx.x] = run;
}

__global__ void __launch_bounds__(256) k_slice_scatter(const uint32_t* __restrict__ part, const uint32_t* __restrict__ block_base,
                                                       uint32_t nbins_total, uint32_t nslices, const uint32_t* __restrict__ slice_base,
                                                       const uint32_t* __restrict__ slicehist, const uint32_t* __restrict__ subbase,
                                                       uint32_t* __restrict__ sorted) {
  __shared__ uint32_t cnt[256];          // this slice's count per sub-bucket, then its exclusive scan (local base)
  __shared__ uint32_t cur[256];          // local cursors
  __shared__ uint32_t gdst[256];         // global destination of the slice's run of each sub-bucket
  __shared__ uint32_t stage[SLICE];
  uint32_t g, s;
  if (!slice_of_block(slice_base, nbins_total, blockIdx.x, g, s)) return;
  const uint32_t start = block_base[(size_t)g * nslices], end = block_base[(size_t)(g + 1) * nslices];
  const uint32_t lo = start + s * SLICE, hi = (lo + SLICE < end) ? lo + SLICE : end;
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t span = ((hi - lo + 255u) / 256u) * 256u;
  for (uint32_t o = threadIdx.x; o < span; o += 256) {
    const bool live = lo + o < hi;
    lds_ranked_inc(cnt, live ? (part[lo + o] >> 24) : 0u, live);
  }
  __syncthreads();
  const uint32_t mine = cnt[threadIdx.x];
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    uint32_t u = ((int)threadIdx.x >= d) ? cnt[threadIdx.x - d] : 0u;
    __syncthreads();
    cnt[threadIdx.x] += u;
    __syncthreads();
  }
  const uint32_t lbase = cnt[threadIdx.x] - mine;
  __syncthreads();
  cnt[threadIdx.x] = lbase;
  cur[threadIdx.x] = lbase;
  gdst[threadIdx.x] = subbase[(size_t)g * 256u + threadIdx.x] + slicehist[(size_t)blockIdx.x * 256u + threadIdx.x];
  __syncthreads();
  for (uint32_t o = threadIdx.x; o < span; o += 256) {
    const bool live = lo + o < hi;
    const uint32_t v = live ? part[lo + o] : 0u;
    const uint32_t pos = lds_ranked_inc(cur, v >> 24, live);
    if (live) stage[pos] = v;
  }
  __syncthreads();
  for (uint32_t o = threadIdx.x; o < hi - lo; o += 256) {           // consecutive o of one sub-bucket -> consecutive addresses
    const uint32_t v = stage[o], k = v >> 24;
    sorted[gdst[k] + (o - cnt[k])] = (v & 0x7fffffu) | ((v >> 23 & 1u) << 31);
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

// The same exclusive scan as ONE launch of one 1024-thread block, for n <= USCAN1_MAX items: each thread scans a contiguous run
// (three launches of ~4.7 us each are all latency at the sizes of a mid-size MSM's sort: 5 K block counts at 2^16 terms).
constexpr uint32_t USCAN1_MAX = 1u << 15;
__global__ void __launch_bounds__(1024) k_uscan_one(uint32_t* __restrict__ a, uint32_t n) {
  __shared__ uint32_t sh[1024];
  const uint32_t per = (n + 1023u) / 1024u, lo = threadIdx.x * per, hi = (lo + per < n) ? lo + per : n;
  uint32_t local = 0;
  for (uint32_t i = lo; i < hi; ++i) local += a[i];
  sh[threadIdx.x] = local;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    uint32_t u = ((int)threadIdx.x >= d) ? sh[threadIdx.x - d] : 0u;
    __syncthreads();
    sh[threadIdx.x] += u;
    __syncthreads();
  }
  uint32_t excl = sh[threadIdx.x] - local;
  for (uint32_t i = lo; i < hi; ++i) { const uint32_t v = a[i]; a[i] = excl; excl += v; }
  if (threadIdx.x == 1023) a[n] = sh[1023];
}

// ------------------------------------------------------------------ regime B: many independent small MSMs
// A batch of M MSMs (MSM j = terms [offs[j], offs[j+1]) of one concatenated input) -- e.g. the 5*ell+7-term
// final MSMs of 1024 MSMAccumulator.verify() calls (msm_accumulator.py:60-68).  Bucket space is indexed by
// group g = j*nwin + w; with NB <= 256 buckets per group the counting sort of a group lives in one block's
// LDS.  Everything between the sort and the bucket sums (chunking, length ordering, k_accumulate, k_seg_reduce)
// is the same code as regime A, so lanes of one wave work on chunks of equal length from ANY msm/window.
// split != 0 (the endomorphism split, glv.h): a row of digits holds 2 * split entries -- the k1 halves of all terms, then the k2
// halves -- so MSM j owns the two runs [o0, o1) and [split + o0, split + o1); N is the row stride.
__global__ void __launch_bounds__(256) k_group_count(const uint16_t* __restrict__ digits, const uint32_t* __restrict__ offs,
                                                     uint32_t* __restrict__ hist, uint32_t N, uint32_t NB, uint32_t nwin, uint32_t split) {
  __shared__ uint32_t cnt[256];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t j = blockIdx.x, w = blockIdx.y;
  const uint32_t o0 = offs[j], o1 = offs[j + 1];
  const uint32_t span = ((o1 - o0 + 255u) / 256u) * 256u;
  for (uint32_t half = 0; half < (split ? 2u : 1u); ++half) {
    const uint16_t* dg = digits + (size_t)w * N + (size_t)half * split;
    for (uint32_t o = threadIdx.x; o < span; o += 256) {
      const uint32_t e = (o0 + o < o1) ? (uint32_t)dg[o0 + o] : 0u;
      uint32_t neg;
      lds_ranked_inc(cnt, e ? digit_mag(e, neg) - 1u : 0u, e != 0u);
    }
  }
  __syncthreads();
  if (threadIdx.x < NB) hist[((size_t)j * nwin + w) * NB + threadIdx.x] = cnt[threadIdx.x];
}

__global__ void __launch_bounds__(256) k_group_scatter(const uint16_t* __restrict__ digits, const uint32_t* __restrict__ offs,
                                                       const uint32_t* __restrict__ off, uint32_t* __restrict__ sorted,
                                                       uint32_t N, uint32_t NB, uint32_t nwin, uint32_t split) {
  __shared__ uint32_t cur[256];
  const uint32_t j = blockIdx.x, w = blockIdx.y;
  if (threadIdx.x < NB) cur[threadIdx.x] = off[((size_t)j * nwin + w) * NB + threadIdx.x];
  __syncthreads();
  const uint32_t o0 = offs[j], o1 = offs[j + 1];
  const uint16_t* dg = digits + (size_t)w * N;
  const uint32_t span = ((o1 - o0 + 255u) / 256u) * 256u;
  for (uint32_t half = 0; half < (split ? 2u : 1u); ++half) {
    for (uint32_t o = threadIdx.x; o < span; o += 256) {
      const uint32_t i = half * split + o0 + o;                 // entry = record index: phi(P_t) is record split + t
      const uint32_t e = (o0 + o < o1) ? (uint32_t)dg[i] : 0u;
      uint32_t neg = 0;
      const uint32_t b = e ? digit_mag(e, neg) - 1u : 0u;
      const uint32_t pos = lds_ranked_inc(cur, b, e != 0u);
      if (e) sorted[pos] = i | (neg << 31);
    }
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

// k_scan1 + k_scan2 + k_scan3 as one launch of one 1024-thread block, for nb_total <= SCAN1_MAX buckets
constexpr uint32_t SCAN1_MAX = 1u << 13;        // beyond a few thousand buckets the serial runs of one block cost more than three launches (170 us at 82 K buckets)
__global__ void __launch_bounds__(1024) k_scan_one(const uint32_t* __restrict__ hist, uint32_t* __restrict__ off, uint32_t* __restrict__ choff,
                                                    uint32_t nb_total, uint32_t L0) {
  __shared__ uint2 sh[1024];
  const uint32_t per = (nb_total + 1023u) / 1024u, lo = threadIdx.x * per, hi = (lo + per < nb_total) ? lo + per : nb_total;
  uint2 local = make_uint2(0, 0);
  for (uint32_t i = lo; i < hi; ++i) { const uint32_t c = hist[i]; local.x += c; local.y += chunk_count(c, L0); }
  sh[threadIdx.x] = local;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    uint2 v = make_uint2(0, 0);
    if ((int)threadIdx.x >= d) v = sh[threadIdx.x - d];
    __syncthreads();
    sh[threadIdx.x].x += v.x; sh[threadIdx.x].y += v.y;
    __syncthreads();
  }
  uint2 excl = make_uint2(sh[threadIdx.x].x - local.x, sh[threadIdx.x].y - local.y);
  for (uint32_t i = lo; i < hi; ++i) {
    const uint32_t c = hist[i];
    off[i] = excl.x; choff[i] = excl.y;
    excl.x += c; excl.y += chunk_count(c, L0);
  }
  if (threadIdx.x == 1023) { off[nb_total] = sh[1023].x; choff[nb_total] = sh[1023].y; }
}

constexpr uint32_t CHUNK_DESC_BLOCKS = 320;
__global__ void __launch_bounds__(256) k_chunk_desc(const uint32_t* __restrict__ off, const uint32_t* __restrict__ choff,
                                                    uint2* __restrict__ desc, uint32_t* __restrict__ len_hist,
                                                    uint32_t* __restrict__ heavy /* [0] = count, then bucket ids */,
                                                    uint32_t heavy_cap, uint32_t nb_total, uint32_t L0, uint32_t* __restrict__ any_multi = nullptr) {
  __shared__ uint32_t sh[LEN_BINS];
  sh[threadIdx.x] = 0;
  __syncthreads();
  // Several tiles of 256 buckets per block when there are many: every block ends with one global atomic per length key it met, and with whole-bucket chunks a block
  // meets ~30 keys -- 2 048 blocks x 30 atomics on 30 addresses were most of this kernel's 30 us at 2^20 terms
#pragma unroll 1
  for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b - threadIdx.x < nb_total; b += gridDim.x * 256) {      // grid-stride: at most CHUNK_DESC_BLOCKS blocks
  if (b < nb_total) {
    uint32_t start = off[b], cnt = off[b + 1] - start;
    if (cnt) {
      uint32_t L = chunk_len(cnt, L0), nch = (cnt + L - 1) / L, o = choff[b];
      for (uint32_t k = 0; k < nch; ++k) {
        uint32_t s = k * L, len = (cnt - s < L) ? (cnt - s) : L;
        desc[o + k] = make_uint2(start + s, len);
      }
      // the histogram of chunk lengths: nch - 1 full chunks and the last one -- two LDS atomics per bucket, not one per chunk (most
      // lanes of a block hit the same key: the atomics serialise)
      if (nch > 1u) atomicAdd(&sh[len_key(L)], nch - 1u);
      atomicAdd(&sh[len_key(cnt - (nch - 1u) * L)], 1u);
      if (nch >= HEAVY_MIN_CHUNKS) {
        uint32_t slot = atomicAdd(&heavy[0], 1u);
        if (slot < heavy_cap) heavy[1 + slot] = b;
      } else if (nch >= 2u && any_multi) {
        *any_multi = 1u;                 // k_bucket_fold has work (benign race: every writer stores 1)
      }
    }
  }
  }
  __syncthreads();
  uint32_t v = sh[threadIdx.x];
  if (v) atomicAdd(&len_hist[threadIdx.x], v);
}

// one block: len_cursor[k] = number of chunks with a LONGER key (descending order => longest chunks first)
// (also copies the call's totals -- sorted entries, chunks -- next to the input-validation flag word for the host)
__global__ void __launch_bounds__(256) k_len_scan(const uint32_t* __restrict__ len_hist, uint32_t* __restrict__ len_cursor,
                                                  const uint32_t* __restrict__ total_entries, const uint32_t* __restrict__ total_chunks,
                                                  uint32_t* __restrict__ stats) {
  __shared__ uint32_t sh[LEN_BINS];
  if (threadIdx.x == 0) { stats[1] = *total_entries; stats[2] = *total_chunks; }
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
// ORDER_PER tiles of 256 chunks per block share ONE LDS histogram and therefore one global atomic per length key: with whole-bucket
// chunks a tile meets ~30 keys, and 2 048 tiles x 30 atomics on 30 addresses were most of this kernel's 31 us at 2^20 terms.
constexpr uint32_t ORDER_PER = 8;
__global__ void __launch_bounds__(256) k_order(const uint2* __restrict__ desc, const uint32_t* __restrict__ total_chunks,
                                               uint32_t* __restrict__ len_cursor, uint32_t* __restrict__ order) {
  __shared__ uint32_t cnt[LEN_BINS];
  __shared__ uint32_t base[LEN_BINS];
  cnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t total = *total_chunks;
  uint32_t key[ORDER_PER], local[ORDER_PER];
#pragma unroll
  for (uint32_t rep = 0; rep < ORDER_PER; ++rep) {
    const uint32_t t = (blockIdx.x * ORDER_PER + rep) * 256 + threadIdx.x;
    const bool live = t < total;
    key[rep] = live ? len_key(desc[t].y) : 0u;
    local[rep] = 0;
    // the lanes that carry the first live lane's key take their slots from one LDS atomic, the others from their own
    const unsigned long long amask = __ballot(live);
    if (amask) {
      const int leader = __ffsll((long long)amask) - 1;
      const uint32_t k0 = __shfl(key[rep], leader, 64);
      const unsigned long long same = __ballot(live && key[rep] == k0);
      const uint32_t lane = threadIdx.x & 63u;
      uint32_t b0 = 0;
      if ((int)lane == leader) b0 = atomicAdd(&cnt[k0], (uint32_t)__popcll(same));
      b0 = __shfl(b0, leader, 64);
      if (live) local[rep] = (key[rep] == k0) ? b0 + (uint32_t)__popcll(same & ((1ull << lane) - 1ull)) : atomicAdd(&cnt[key[rep]], 1u);
    }
  }
  __syncthreads();
  const uint32_t c = cnt[threadIdx.x];
  if (c) base[threadIdx.x] = atomicAdd(&len_cursor[threadIdx.x], c);
  __syncthreads();
#pragma unroll
  for (uint32_t rep = 0; rep < ORDER_PER; ++rep) {
    const uint32_t t = (blockIdx.x * ORDER_PER + rep) * 256 + threadIdx.x;
    if (t < total) order[base[key[rep]] + local[rep]] = t;
  }
}

