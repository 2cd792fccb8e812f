// Bucket reductions: running-sum segments + bit tree (A/B fallback, regime B) and the 2-D row/column reduction.
// Part of the single translation unit csrc/msm_gpu.hip (included inside namespace cg1).
#pragma once

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

// Shuffle tree over the `lanes` partial sums of one row / column (lane `part` of the group starting at wave lane
// `base`).  The levels that have at most lanes/4 additions left run as QUAD-lane additions (g1_quad.h): the 4 lanes of
// quad v fetch both operands and share the field multiplications (4.5 instead of 14.5 multiply-times per level).
// The result ends in the group's first lane.
__device__ __forceinline__ xyzz group_tree(xyzz acc, uint32_t lanes, uint32_t part, uint32_t base, int use_quad) {
  uint32_t delta = lanes >> 1;
  for (; delta >= 1 && (!use_quad || delta * 4u > lanes); delta >>= 1) {
    xyzz o = shfl_down_xyzz(acc, (int)delta);
    if (part < delta) acc = xyzz_add(acc, o);
  }
  uint32_t stride = 1;                                   // value v is held by lane base + v*stride (all 4 lanes of quad v once stride == 4)
  const uint32_t q = part & 3u, quad = part >> 2;
  for (; delta >= 1; delta >>= 1) {
    const uint32_t v = quad < delta ? quad : 0u;          // idle quads recompute pair 0 (same control flow, result unused)
    const xyzz a = shfl_xyzz(acc, (int)(base + v * stride));
    const xyzz b = shfl_xyzz(acc, (int)(base + (v + delta) * stride));
    acc = quad_add(a, b, q);
    stride = 4;
  }
  return acc;
}

__global__ void __launch_bounds__(256, 2) k_rowcol(const uint32_t* __restrict__ choff, const PointSum* __restrict__ sums,
                                                const uint8_t* __restrict__ combined, PointSum* __restrict__ rowsum,
                                                PointSum* __restrict__ colsum, uint32_t nlw, uint32_t hb, uint32_t lb,
                                                uint32_t nrow_blocks, int use_quad) {
  const uint32_t R = 1u << hb, Cn = 1u << lb;
  const uint32_t lane = threadIdx.x & 63u;
  if (blockIdx.x < nrow_blocks) {
    const uint32_t lpr = Cn < 32u ? Cn : 32u, serial = Cn / lpr;
    const uint32_t gr = blockIdx.x * (256u / lpr) + threadIdx.x / lpr;     // global row = lw * R + h
    const uint32_t part = threadIdx.x % lpr;
    const bool live = gr < nlw * R;
    xyzz acc = xyzz_identity();
    if (live)
      for (uint32_t t = 0; t < serial; ++t) acc = xyzz_add(acc, bucket_sum(choff, sums, combined, gr * Cn + part * serial + t));
    acc = group_tree(acc, lpr, part, lane - part, use_quad && lpr >= 4u);
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
    acc = group_tree(acc, lpc, part, lane - part, use_quad && lpc >= 4u);
    if (live && part == 0) store_sum(colsum + gc, acc);
  }
}

// Quad-lane k_bucket_fold for small bucket counts (same regime as k_rowcol_quad): one DPP quad per bucket adds the bucket's
// 2..16 chunk sums into its first slot (balanced plans of mid-size inputs cut every bucket into a few short chunks so that
// the madd chains of k_accumulate stay short; a lone lane needs ~26 us per addition, a quad ~10).  (Several buckets per quad, one after
// the other, to have fewer resident waves -- what helped k_rowcol_quad -- was measured a loss here at every size: 0.223 -> 0.249 ms at 2^16.)
__global__ void __launch_bounds__(256, 2) k_bucket_fold_quad(const uint32_t* __restrict__ choff, PointSum* __restrict__ sums,
                                                             uint8_t* __restrict__ combined, uint32_t nb_total,
                                                             const uint32_t* __restrict__ any_multi) {
  if (*any_multi == 0u) return;
  const uint32_t t = blockIdx.x * 256 + threadIdx.x, b = t >> 2, q = t & 3u;
  if (b >= nb_total) return;                               // whole quads leave together
  const uint32_t c0 = choff[b], c1 = choff[b + 1];
  if (c1 - c0 < 2u || c1 - c0 >= HEAVY_MIN_CHUNKS) return;
  xyzz acc = load_sum(sums + c0);
#pragma unroll 1
  for (uint32_t k = c0 + 1; k < c1; ++k) acc = quad_add(acc, load_sum(sums + k), q);
  if (q == 0) { store_sum(sums + c0, acc); combined[b] = 1; }
}

// Quad-lane variant of k_rowcol for SMALL bucket counts (balanced plans of mid-size inputs: <= 2^18 buckets), where the
// reduction is bound by its chain of dependent EC additions (~26 us each in one lane of a lone wave), not by throughput:
// 2^lgq DPP quads per row / column (16, 8 or 4: a wave carries 1, 2 or 4 rows), quad p takes the elements p, p + 2^lgq, ... (each
// addition shared by the 4 lanes of a quad, g1_quad.h: ~2.6x shorter), then lgq quad-shuffle levels.  Same results as k_rowcol.
// The host picks lgq so that (rounds of resident waves) x (steps per wave) is smallest: the kernel keeps 2 waves per SIMD = 2 048 on
// the chip, and a 2^16-term call has 2 560 rows + columns -- one wave each was 1.25 rounds of 8 steps (144 us), half a wave each is one
// round of 11 (profiles/r04_gap_trace.txt).
__global__ void __launch_bounds__(256, 2) k_rowcol_quad(const uint32_t* __restrict__ choff, const PointSum* __restrict__ sums,
                                                        PointSum* __restrict__ rowsum, PointSum* __restrict__ colsum,
                                                        uint32_t nlw, uint32_t hb, uint32_t lb, uint32_t lgq) {
  const uint32_t R = 1u << hb, Cn = 1u << lb, G = 1u << lgq, per_wave = 16u >> lgq;
  const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u, q = lane & 3u, quad = lane >> 2;
  const uint32_t gw = wave * per_wave + (quad >> lgq), p = quad & (G - 1u);     // this quad's row / column, its place among the row's quads
  const uint32_t nrows = nlw * R, ncols = nlw * Cn;
  if (wave * per_wave >= nrows + ncols) return;                                 // whole waves leave together
  const bool live_row = gw < nrows + ncols;
  // element i of this row / column is bucket first + i * stride, i < len (a wave may carry rows AND columns: the step count below is
  // the longer one's, the bounds are each quad's own)
  const bool is_row = gw < nrows;
  uint32_t first = 0, stride = 1u;
  const uint32_t len = is_row ? Cn : R;
  if (live_row) {
    if (is_row) { first = gw * Cn; stride = 1u; }
    else { const uint32_t gc = gw - nrows, lw = gc / Cn, l = gc % Cn; first = lw * R * Cn + l; stride = Cn; }
  }
  const uint32_t nser = ((R > Cn ? R : Cn) + G - 1u) >> lgq;   // serial steps per quad (wave-uniform)
  xyzz acc = xyzz_identity();
  // ONE quad_add call site for the serial steps and the shuffle levels (the kernel must stay inside the instruction
  // cache and under 256 registers).  k_bucket_fold / k_heavy_combine ran before: a non-empty bucket's sum is its first slot.
#pragma unroll 1
  for (uint32_t step = 0; step < nser + lgq; ++step) {
    xyzz o;
    bool go;
    if (step < nser) {
      const uint32_t i = p + G * step;
      const uint32_t b = first + (i < len ? i : 0u) * stride;
      const uint32_t c0 = choff[b];
      go = live_row && i < len && choff[b + 1] > c0;
      o = load_sum(sums + c0);                               // (a valid address even for an empty bucket: the next bucket's slot or the pad)
    } else {
      const uint32_t dq = (G >> 1) >> (step - nser);
      o = shfl_down_xyzz(acc, (int)(4u * dq));
      go = p < dq;
    }
    if (go) acc = quad_add(acc, o, q);
  }
  if (live_row && p == 0u && q == 0u) store_sum(is_row ? rowsum + gw : colsum + (gw - nrows), acc);
}

// k_rowcol_quad with the cross-quad levels on rows: ONE WAVE per row / column, its 16 quads take the elements p, p + 16, ... (len / 16
// serial quad additions), park their partial sums in LDS, and the wave then adds the 16 partials one after the other with one limb per
// lane (fp_row.h: ~2.4 us per addition where a quad-shuffle level costs ~23: 56 words through ds_bpermute plus a quad addition).
// 64-element rows: 4 + 15 short steps instead of 16 + 2 long ones.  The sums are written in row form (nearly normal limbs): only
// k_small_tree_row reads them.
__global__ void __launch_bounds__(256, 2) k_rowcol_quad_row(const uint32_t* __restrict__ choff, const PointSum* __restrict__ sums,
                                                            const uint8_t* __restrict__ combined,
                                                            PointSum* __restrict__ rowsum, PointSum* __restrict__ colsum,
                                                            uint32_t nlw, uint32_t hb, uint32_t lb) {
  __shared__ PointSum part[4 * 16];
  const uint32_t R = 1u << hb, Cn = 1u << lb;
  const uint32_t wib = threadIdx.x >> 6, gw = blockIdx.x * 4u + wib, lane = threadIdx.x & 63u, q = lane & 3u, p = lane >> 2;
  const uint32_t nrows = nlw * R, ncols = nlw * Cn;
  if (gw >= nrows + ncols) return;                                              // whole waves leave together (no block-wide barrier below)
  const bool is_row = gw < nrows;
  uint32_t first, stride;
  const uint32_t len = is_row ? Cn : R;
  if (is_row) { first = gw * Cn; stride = 1u; }
  else { const uint32_t gc = gw - nrows, lw = gc / Cn, l = gc % Cn; first = lw * R * Cn + l; stride = Cn; }
  // The quad walks its elements p, p + 16, ... and, inside an element (a bucket), the bucket's chunk sums: a bucket cut into a few chunks
  // needs no k_bucket_fold_quad pass in front (75 us at 2^16 terms for one addition per bucket); buckets k_heavy_combine joined have
  // their total in the first slot.  ONE quad_add call site.
  xyzz acc = xyzz_identity();
  uint32_t i = p, c0 = 0, nch = 0, kk = 0;
  bool fresh = true;
#pragma unroll 1
  for (;;) {
    if (fresh) {
      if (i >= len) break;
      const uint32_t b = first + i * stride;
      c0 = choff[b];
      nch = choff[b + 1] - c0;
      if (nch > 1u && combined[b]) nch = 1u;
      kk = 0; i += 16u; fresh = false;
    }
    if (kk < nch) { acc = quad_add(acc, load_sum(sums + c0 + kk), q); ++kk; }
    if (kk >= nch) fresh = true;
  }
  if (q == 0u) store_sum(&part[wib * 16u + p], acc);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);                                           // lgkmcnt(0): the wave's own LDS stores have landed
  const RowK k = row_constants();
  xyzz_row racc = row_load_sum(&part[wib * 16u], k.lane16);
  const uint32_t np = len < 16u ? len : 16u;                                    // quads beyond the row's length hold the identity
#pragma unroll 1
  for (uint32_t j = 1; j < np; ++j) racc = row_add(racc, row_load_sum(&part[wib * 16u + j], k.lane16), k);
  row_store_sum(is_row ? rowsum + gw : colsum + (gw - nrows), racc, k.lane16);
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


// XYZZ partial sums -> canonical words in the host's Montgomery form (regime B with few MSMs: their Horner runs on the host)
__global__ void __launch_bounds__(64) k_export_sums(const PointSum* __restrict__ src, PointWords* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  xyzz_words o;
  xyzz_export(load_sum(src + i), o);
  PointWords* dst = out + i;
  for (int cidx = 0; cidx < 4; ++cidx) for (int k = 0; k < 12; ++k) dst->w[cidx][k] = o.w[cidx][k];
  dst->inf = o.inf;
}

// The last kernel of an MSM call: the exported window sums (+ the status words behind them) from device memory into MAPPED HOST
// memory, 16 bytes per store, then -- after a system-scope fence -- the call's sequence number into the flag word the host polls.
// One block: 66 KB at most.
__global__ void __launch_bounds__(1024) k_export_host(const uint4* __restrict__ src, uint4* __restrict__ dst_host, uint32_t nvec,
                                                      uint32_t* __restrict__ flag_host, uint32_t seq) {
  for (uint32_t i = threadIdx.x; i < nvec; i += 1024) dst_host[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ------------------------------------------------------------------ quad-lane variants (g1_quad.h)
// Same results as k_small_tree / k_msm_horner; every EC operation is shared by the 4 lanes of a DPP quad, which
// shortens each dependent step ~2.5x.  These kernels are purely latency-bound (a few hundred waves at most).

// grid = (1 + hb + lb, nlw); blockDim = 64 .. 512 threads = NQ quads (the host sizes it to the longer of the two sums): each quad
// takes elements Q, Q + NQ, ... of its item, then a quad tree inside each wave and one across the block's waves.
__global__ void __launch_bounds__(512) k_small_tree_quad(const PointSum* __restrict__ rowsum, const PointSum* __restrict__ colsum,
                                                         PointWords* __restrict__ out, uint32_t hb, uint32_t lb) {
  __shared__ PointSum sh[8];
  const uint32_t item = blockIdx.x, lw = blockIdx.y;
  const uint32_t q = threadIdx.x & 3u, Q = threadIdx.x >> 2, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t NQ = blockDim.x >> 2, NW = blockDim.x >> 6;
  const bool on_rows = item <= hb;
  const uint32_t J = on_rows ? (1u << hb) : (1u << lb);
  const PointSum* src = on_rows ? rowsum + (size_t)lw * J : colsum + (size_t)lw * J;
  const uint32_t bit = on_rows ? item - 1u : item - 1u - hb;
  xyzz acc = xyzz_identity();
  for (uint32_t e = Q; e < J; e += NQ)
    if (item == 0 || ((e >> bit) & 1u)) acc = quad_add(acc, load_sum(src + e), q);
  for (uint32_t dq = 8; dq >= 1; dq >>= 1) {               // 16 quads per wave
    xyzz o = shfl_down_xyzz(acc, (int)(4u * dq));
    if ((lane >> 2) < dq) acc = quad_add(acc, o, q);
  }
  if (NW > 1) {
    if (lane == 0) store_sum(&sh[wave], acc);
    __syncthreads();
    if (wave == 0) {
      acc = (Q < NW) ? load_sum(&sh[Q]) : xyzz_identity();
      for (uint32_t dq = NW >> 1; dq >= 1; dq >>= 1) {
        xyzz o = shfl_down_xyzz(acc, (int)(4u * dq));
        if (Q < dq) acc = quad_add(acc, o, q);
      }
    }
  }
  if (threadIdx.x == 0) {
    xyzz_words o;
    xyzz_export(acc, o);
    PointWords* dst = out + (size_t)lw * gridDim.x + item;
    for (int cidx = 0; cidx < 4; ++cidx) for (int k = 0; k < 12; ++k) dst->w[cidx][k] = o.w[cidx][k];
    dst->inf = o.inf;
  }
}

// one QUAD per MSM
__global__ void __launch_bounds__(64) k_msm_horner_quad(const PointSum* __restrict__ group_sum, PointWords* __restrict__ out,
                                                        uint32_t M, uint32_t nwin, uint32_t c) {
  const uint32_t t = blockIdx.x * 64 + threadIdx.x, j = t >> 2, q = t & 3u;
  if (j >= M) return;                                       // whole quads leave together
  xyzz acc = xyzz_identity();
  for (int w = (int)nwin - 1; w >= 0; --w) {
    for (uint32_t k = 0; k < c; ++k) acc = quad_dbl(acc, q);
    acc = quad_add(acc, load_sum(group_sum + (size_t)j * nwin + w), q);
  }
  if (q == 0) {
    xyzz_words o;
    xyzz_export(acc, o);
    PointWords* dst = out + j;
    for (int cidx = 0; cidx < 4; ++cidx) for (int k = 0; k < 12; ++k) dst->w[cidx][k] = o.w[cidx][k];
    dst->inf = o.inf;
  }
}
