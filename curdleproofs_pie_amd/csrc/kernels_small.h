// k_msm_small: ONE launch for a whole MSM of n <= 1024 terms -- the sizes the protocol itself issues (4 ... 627 terms:
// ipa.py:223,232; same_msm.py:219-226; msm_accumulator.py:64).  The regime-A chain costs ~20 launches there, every one of them a
// latency hop (>= 5 us of launch + drain each around kernels that keep a few waves busy).
// Part of the single translation unit csrc/msm_gpu.hip (included inside namespace cg1).
//
// Grid = (windows, point slices of <= 256 terms, independent MSMs of one call); 512 threads = 128 DPP quads per workgroup, everything between the scalar words and
// the exported window items stays in LDS (111 KB):
//   digits      one thread per term: signed c-bit digit of ITS window (recoding carry walked up from window 0), the point converted
//               to Montgomery limbs into LDS (affine96 / normalised blob input; prepared records are copied)
//   sort        LDS counting sort by bucket (histogram with returning atomics, one packed scan for bucket and chunk offsets)
//   accumulate  buckets cut into chunks of <= 4 entries, one quad per chunk (g1_quad.h additions): uniform scalars give one short
//               chain per bucket, the all-equal-scalars pattern of the callers (every term in ONE bucket) gives n / 4 chunks ...
//   fold        ... that a pairwise tree per bucket joins in log2(chunks) levels
//   row / col   2-D bucket reduction (kernels_reduce.h): 2^hb row sums + 2^lb column sums, 4 quads each
//   items       1 + hb + lb masked sums per window, 8 quads each -- the points whose power-of-two weights the host Horner applies
//   combine     with several slices per window the last workgroup to arrive (ticket) adds their items (a tree over pairs of slices); the last window to finish
//               publishes the status words and the call's sequence number into mapped host memory (zero-copy export)
// All EC additions of all phases go through ONE quad_add call site inside a phase state machine: the kernel has to stay inside the
// instruction cache and under 256 registers.
#pragma once

constexpr uint32_t SM_MAX_N = 2048;       // terms per call (the combine step is a tree over up to 8 pairs of slices: S <= 16)
constexpr uint32_t SM_SLICE = 256;        // terms per workgroup
constexpr uint32_t SM_L = 4;              // entries per chunk
constexpr uint32_t SM_QUADS = 128;

constexpr uint32_t SM_MAX_MSMS = 64;      // independent MSMs one launch may carry (grid.z)
constexpr uint32_t SM_ONE_ROUND = 256;    // workgroups of ONE MSM: windows x slices within one round of the chip's CUs (a workgroup fills a CU)
constexpr uint32_t SM_MAX_GROUPS = 2560;  // ... as long as windows x slices x MSMs stays a few rounds of workgroups (no workgroup waits for another: any number is safe)

struct SmallArgs {
  const void* src;                        // SRC 0: n x affine96; 1: n x point blob with Z in {0, 1}; 2: PreparedPoint records
  const uint8_t* flags;                   // SRC 2: identity flags
  const uint32_t* scalars;
  const uint32_t* offs;                   // M + 1 term offsets of the M MSMs inside src / scalars (device); NULL: one MSM of n terms
  uint32_t n, M, S, c, nwin, hb, lb, nitems;
  PointSum* partial;                      // [M][nwin][S][nitems]  (S > 1)
  uint32_t* counters;                     // [0, M nwin): window tickets; then: finished windows | bad-scalar flag | entries.  Zero between calls.
  PointWords* out_host;                   // mapped host memory: M x nwin x nitems records, then one record of status words
  uint32_t* flag_host;
  uint32_t seq;
  uint32_t row_tail;                      // 1: the items, the combine and the export run one limb per lane, one wave per item (fp_row.h)
  uint32_t glv;                           // 1: the endomorphism split (glv.h; the caller vouches for G1): an MSM of n terms runs as 2n entries -- entry v < n is
                                          // (k1 of scalar v, P_v), entry n + v is (k2, phi(P_v)) -- over the windows of a 128-position plan; S counts slices of 2n
};

template <int SRC>
__global__ void __launch_bounds__(512) k_msm_small(SmallArgs a) {
  __shared__ PointSum s_sum[SM_SLICE];
  __shared__ PreparedPoint s_pts[SM_SLICE];
  __shared__ PointSum s_rc[32];                              // row sums [0, 16), column sums [16, 32)
  __shared__ uint32_t s_hist[256], s_off[260], s_choff[260], s_sorted[256], s_wtot[4], s_misc[8];
  __shared__ uint16_t s_cstart[256], s_cbucket[256];
  __shared__ uint8_t s_clen[256];

  const uint32_t tid = threadIdx.x, w = blockIdx.x, sl = blockIdx.y, msm = blockIdx.z;
  const uint32_t c = a.c, NB = 1u << (c - 1), nwin = a.nwin, S = a.S, nitems = a.nitems;
  const uint32_t first = a.offs ? a.offs[msm] : 0u, n_msm = a.offs ? a.offs[msm + 1] - first : a.n;
  const uint32_t base = sl * SM_SLICE;                          // (an MSM shorter than the launch's longest leaves its last slices empty)
  const uint32_t n_ent = a.glv ? 2u * n_msm : n_msm;           // entries of this MSM: terms, or both halves of every term
  const uint32_t ns = base >= n_ent ? 0u : ((n_ent - base < SM_SLICE) ? n_ent - base : SM_SLICE);
  const uint32_t wslot = msm * nwin + w, gctr = a.M * nwin;     // this (MSM, window)'s slot; the launch-wide counters behind the tickets
  if (tid < 256) s_hist[tid] = 0;
  if (tid < 8) s_misc[tid] = 0;
  __syncthreads();

  // ---- digits + points
  uint32_t my_b = 0, my_pos = 0, my_neg = 0;
  bool my_valid = false;
  if (tid < ns) {
    const uint32_t ve = base + tid;
    const bool second = a.glv && ve >= n_msm;                    // the phi(P) half of a split term
    const uint32_t i = first + (second ? ve - n_msm : ve);
    uint32_t inf;
    fp x, y;
    if (SRC == 0) {
      const uint4* r = reinterpret_cast<const uint4*>(static_cast<const uint32_t*>(a.src) + 24ull * i);
      uint32_t wd[24], any = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { const uint4 v = r[k]; wd[4 * k] = v.x; wd[4 * k + 1] = v.y; wd[4 * k + 2] = v.z; wd[4 * k + 3] = v.w; }
#pragma unroll
      for (int k = 0; k < 24; ++k) any |= wd[k];
      x = fp_to_mont(fp_from_words(wd));
      y = fp_to_mont(fp_from_words(wd + 12));
      inf = any ? 0u : 1u;
    } else if (SRC == 1) {
      const uint32_t* r = static_cast<const uint32_t*>(a.src) + 36ull * i;
      uint32_t wd[12], zany = 0;
      load_words12(r + 24, wd);
#pragma unroll
      for (int k = 0; k < 12; ++k) zany |= wd[k];
      load_words12(r, wd);
      x = fp_from_host_words(wd);
      load_words12(r + 12, wd);
      y = fp_from_host_words(wd);
      inf = zany ? 0u : 1u;
    } else {
      uint32_t fl;
      load_affine(static_cast<const PreparedPoint*>(a.src) + i, x, y, fl);
      inf = a.flags[i];
    }
    if (second) {
      constexpr uint32_t bt[NL] = {D_BETA[0], D_BETA[1], D_BETA[2], D_BETA[3], D_BETA[4], D_BETA[5], D_BETA[6], D_BETA[7], D_BETA[8], D_BETA[9], D_BETA[10], D_BETA[11], D_BETA[12], D_BETA[13]};
      fp beta; for (int k = 0; k < NL; ++k) beta.l[k] = bt[k];
      x = fp_norm(fp_mul(x, beta));
    }
    uint32_t* o = reinterpret_cast<uint32_t*>(&s_pts[tid]);
#pragma unroll
    for (int k = 0; k < NL; ++k) { o[k] = x.l[k]; o[NL + k] = y.l[k]; }
    o[2 * NL] = inf;
    DigitIter it;
    load_scalar(a.scalars, i, it);
    if (it.s[7] >> 31) atomicOr(&a.counters[gctr + 1], 1u);        // a scalar >= 2^255: the host rejects the call
    if (a.glv) {
      GlvParts g;
      glv_split(it.s, g);
      if (second) load_half(g.k2, g.neg2, it); else load_half(g.k1, g.neg1, it);
    }
    WinPlan pl;
    pl.nwin = (int)nwin; pl.cmax = (int)c; pl.n_hi = (int)nwin; pl.glv = 0;
    int d = 0;
    for (uint32_t ww = 0; ww <= w; ++ww) d = it.next(pl, (int)ww);
    if (!inf && d != 0) {
      my_valid = true;
      my_neg = d < 0 ? 1u : 0u;
      my_b = (uint32_t)(d < 0 ? -d : d) - 1u;
      my_pos = atomicAdd(&s_hist[my_b], 1u);
    }
  }
  __syncthreads();

  // ---- one packed scan: bucket offsets (low half) and chunk offsets (high half); both totals <= 256
  uint32_t v = 0, incl = 0;
  if (tid < 256) {
    const uint32_t cnt = tid < NB ? s_hist[tid] : 0u;
    const uint32_t nch = (cnt + SM_L - 1) / SM_L;
    v = cnt | (nch << 16);
    incl = v;
    const uint32_t ln = tid & 63u;
    for (int dlt = 1; dlt < 64; dlt <<= 1) { const uint32_t u = __shfl_up(incl, dlt, 64); if (ln >= (uint32_t)dlt) incl += u; }
    if (ln == 63u) s_wtot[tid >> 6] = incl;
    atomicMax(&s_misc[0], nch);
  }
  __syncthreads();
  if (tid < 256) {
    for (uint32_t k = 0; k < (tid >> 6); ++k) incl += s_wtot[k];
    const uint32_t excl = incl - v;
    s_off[tid] = excl & 0xffffu;
    s_choff[tid] = excl >> 16;
    if (tid == 255) { s_off[256] = incl & 0xffffu; s_choff[256] = incl >> 16; }
  }
  __syncthreads();
  if (my_valid) s_sorted[s_off[my_b] + my_pos] = tid | (my_neg << 31);
  if (tid < NB) {
    const uint32_t cnt = s_hist[tid], c0 = s_choff[tid], o0 = s_off[tid];
    for (uint32_t k = 0; k * SM_L < cnt; ++k) {
      s_cstart[c0 + k] = (uint16_t)(o0 + k * SM_L);
      s_clen[c0 + k] = (uint8_t)((cnt - k * SM_L < SM_L) ? cnt - k * SM_L : SM_L);
      s_cbucket[c0 + k] = (uint16_t)tid;
    }
  }
  __syncthreads();

  // ---- the phase machine (uniform control flow; `go` is uniform inside every quad)
  enum : uint32_t { PH_ACC = 0, PH_FOLD, PH_ROWCOL, PH_TREE, PH_COMBINE, PH_EXPORT };
  const uint32_t Q = tid >> 2, q = tid & 3u;
  const uint32_t R = 1u << a.hb, Cn = 1u << a.lb, nsum = R + Cn;
  const uint32_t nchunks = s_choff[256], rounds = (nchunks + SM_QUADS - 1) / SM_QUADS, maxnch = s_misc[0];
  const uint32_t SR = ((R > Cn ? R : Cn) + 3u) / 4u;              // serial steps of a row / column sum on 4 quads
  const uint32_t ST = ((R > Cn ? R : Cn) + 7u) / 8u;              // serial steps of an item's masked sum on 8 quads
  uint32_t comb_dq0 = 1, comb_levels = 0;                         // combine: pairs of slices, then log2 levels over the pairs
  while (comb_dq0 < (S + 1u) / 2u) comb_dq0 <<= 1;
  comb_dq0 >>= 1;
  for (uint32_t d = comb_dq0; d; d >>= 1) ++comb_levels;
  xyzz acc = xyzz_identity();
  uint32_t phase = rounds ? PH_ACC : PH_ROWCOL, i0 = 0, i1 = 0, fold_st = 1;
  bool exp_live = false;
  uint32_t exp_item = 0;
#pragma unroll 1
  for (;;) {
    xyzz o = xyzz_identity();
    bool go = false, from_pts = false;
    const PointSum* lp = nullptr;
    uint32_t shdelta = 0, pt_entry = 0;
    if (phase == PH_ACC) {
      const uint32_t ch = i0 * SM_QUADS + Q;
      const bool live = ch < nchunks;
      const uint32_t len = live ? s_clen[ch] : 0u;
      go = i1 < len;
      pt_entry = s_sorted[go ? s_cstart[ch] + i1 : 0u];
      from_pts = true;
      if (i1 == 0) acc = xyzz_identity();
    } else if (phase == PH_FOLD) {
      const uint32_t ch = i0 * SM_QUADS + Q;
      const bool live = ch < nchunks;
      const uint32_t b = s_cbucket[live ? ch : 0u];
      const uint32_t r = ch - s_choff[b], nb = s_choff[b + 1] - s_choff[b];
      go = live && !(r & (2u * fold_st - 1u)) && (r + fold_st < nb);
      lp = &s_sum[go ? ch + fold_st : 0u];
      acc = load_sum(&s_sum[live ? ch : 0u]);
    } else if (phase == PH_ROWCOL) {
      const uint32_t sg = Q >> 2, part = Q & 3u;
      const bool live = sg < nsum, is_row = sg < R;
      const uint32_t len = is_row ? Cn : R;
      if (i1 == 0) acc = xyzz_identity();
      if (i1 < SR) {
        const uint32_t i = part + 4u * i1;
        const bool in = live && i < len;
        const uint32_t b = in ? (is_row ? sg * Cn + i : i * Cn + (sg - R)) : 0u;
        go = in && s_hist[b] > 0u;
        lp = &s_sum[go ? s_choff[b] : 0u];
      } else {
        const uint32_t dq = 2u >> (i1 - SR);
        shdelta = 4u * dq;
        go = live && part < dq;
      }
    } else if (phase == PH_TREE) {
      const uint32_t item = Q >> 3, part = Q & 7u;
      const bool live = item < nitems, on_rows = item <= a.hb;
      const uint32_t J = on_rows ? R : Cn, bit = on_rows ? item - 1u : item - 1u - a.hb;
      if (i1 == 0) acc = xyzz_identity();
      if (i1 < ST) {
        const uint32_t e = part + 8u * i1;
        go = live && e < J && (item == 0u || ((e >> bit) & 1u));
        lp = &s_rc[(on_rows ? 0u : 16u) + (go ? e : 0u)];
      } else {
        const uint32_t dq = 4u >> (i1 - ST);
        shdelta = 4u * dq;
        go = live && part < dq;
      }
    } else if (phase == PH_COMBINE) {
      // eight quads per item: quad `part` adds slices 2 part and 2 part + 1, then a shuffle tree over the pairs (S <= 16)
      const uint32_t item = Q >> 3, part = Q & 7u;
      const bool live = item < nitems;
      if (i1 == 0) {
        const PointSum* b = a.partial + ((size_t)wslot * S) * nitems + (live ? item : 0u);
        const uint32_t s0 = 2u * part, s1 = s0 + 1u;
        acc = (live && s0 < S) ? load_sum(b + (size_t)s0 * nitems) : xyzz_identity();
        go = live && s1 < S;
        lp = b + (size_t)(go ? s1 : 0u) * nitems;
      } else {
        const uint32_t dq = comb_dq0 >> (i1 - 1u);
        shdelta = 4u * dq;
        go = live && part < dq;
      }
    }
    if (from_pts) {
      fp x, y; uint32_t fl;
      load_affine(&s_pts[pt_entry & 0xffffu], x, y, fl);
      if (pt_entry >> 31) y = fp_neg<3>(y);
      o = xyzz_from_affine(x, y);
    } else if (shdelta) {
      o = shfl_down_xyzz(acc, (int)shdelta);
    } else if (lp) {
      o = load_sum(lp);
    }
    if (go) acc = quad_add(acc, o, q);

    // ---- store / advance
    if (phase == PH_ACC) {
      if (i1 + 1u == SM_L) {
        const uint32_t ch = i0 * SM_QUADS + Q;
        if (ch < nchunks && q == 0u) store_sum(&s_sum[ch], acc);
        i1 = 0;
        if (++i0 == rounds) { i0 = 0; phase = maxnch > 1u ? PH_FOLD : PH_ROWCOL; __syncthreads(); }
      } else {
        ++i1;
      }
    } else if (phase == PH_FOLD) {
      if (go && q == 0u) store_sum(&s_sum[i0 * SM_QUADS + Q], acc);
      if (++i0 == rounds) {
        i0 = 0; fold_st <<= 1;
        __syncthreads();
        if (fold_st >= maxnch) phase = PH_ROWCOL;
      }
    } else if (phase == PH_ROWCOL) {
      if (++i1 == SR + 2u) {
        const uint32_t sg = Q >> 2;
        if (sg < nsum && (Q & 3u) == 0u && q == 0u) store_sum(&s_rc[sg < R ? sg : 16u + (sg - R)], acc);
        i1 = 0; phase = PH_TREE;
        __syncthreads();
        if (a.row_tail && nitems <= 8u) break;                 // the rest runs below, one wave per item
      }
    } else if (phase == PH_TREE) {
      if (++i1 == ST + 3u) {
        const uint32_t item = Q >> 3;
        const bool mine = item < nitems && (Q & 7u) == 0u;
        if (S == 1u) { exp_live = mine; exp_item = item; phase = PH_EXPORT; break; }
        if (mine && q == 0u) store_sum(a.partial + ((size_t)wslot * S + sl) * nitems + item, acc);
        if (tid == 0) atomicAdd(&a.counters[gctr + 2], s_off[256]);
        __threadfence();
        __syncthreads();
        if (tid == 0) s_misc[2] = atomicAdd(&a.counters[wslot], 1u);
        __syncthreads();
        if (s_misc[2] != S - 1u) return;                       // not the last slice of this window
        __threadfence();
        i1 = 0; phase = PH_COMBINE;
      }
    } else {  // PH_COMBINE
      if (++i1 == 1u + comb_levels) { exp_live = (Q >> 3) < nitems && (Q & 7u) == 0u; exp_item = Q >> 3; phase = PH_EXPORT; break; }
    }
  }

  // ---- row tail (a.row_tail): the 1 + hb + lb items of the window, the combine over the slices and the export with ONE WAVE PER ITEM and
  // one limb per lane.  At this point at most nitems * 8 additions are left and they form chains: a quad addition costs ~10 us at this
  // kernel's two waves per SIMD, a wave of rows ~2.4 us -- 7 dependent row additions (item 0) against 4 quad steps, S - 1 against
  // 1 + log2(S) for the combine (profiles/r05_rowlane_ab.txt).
  if (phase == PH_TREE) {
    const RowK k = row_constants();
    const uint32_t wv = tid >> 6;
    const bool live = wv < nitems;
    xyzz_row racc; racc.X = racc.Y = racc.ZZ = racc.ZZZ = 0; racc.inf = 1;
    if (live) {
      const bool on_rows = wv <= a.hb;
      const uint32_t J = on_rows ? R : Cn, bit = on_rows ? wv - 1u : wv - 1u - a.hb;
      for (uint32_t e = 0; e < J; ++e)
        if (wv == 0u || ((e >> bit) & 1u)) racc = row_add(racc, row_load_sum(&s_rc[(on_rows ? 0u : 16u) + e], k.lane16), k);
    }
    if (S > 1u) {
      if (live) row_store_sum(a.partial + ((size_t)wslot * S + sl) * nitems + wv, racc, k.lane16);
      if (tid == 0) atomicAdd(&a.counters[gctr + 2], s_off[256]);
      __threadfence();
      __syncthreads();
      if (tid == 0) s_misc[2] = atomicAdd(&a.counters[wslot], 1u);
      __syncthreads();
      if (s_misc[2] != S - 1u) return;                           // not the last slice of this window
      __threadfence();
      if (live) {
        racc.X = racc.Y = racc.ZZ = racc.ZZZ = 0; racc.inf = 1;
        for (uint32_t s2 = 0; s2 < S; ++s2) racc = row_add(racc, row_load_sum(a.partial + ((size_t)wslot * S + s2) * nitems + wv, k.lane16), k);
      }
    }
    acc = row_to_xyzz(racc, k.lane16);
    exp_live = live && (tid & 63u) < 4u;                         // lanes 0..3 of the item's wave write its four coordinates
    exp_item = wv;
  }

  // ---- export: lane q of the item's quad converts and writes coordinate q (canonical, the host's Montgomery form)
  if (exp_live) {
    PointWords* dst = a.out_host + (size_t)wslot * nitems + exp_item;
    fp coord;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const uint32_t lo = (q & 1u) ? acc.Y.l[k] : acc.X.l[k], hi = (q & 1u) ? acc.ZZZ.l[k] : acc.ZZ.l[k];
      coord.l[k] = (q & 2u) ? hi : lo;
    }
    uint32_t ow[12];
    fp_to_host_words(coord, ow);
    if (acc.inf) {
#pragma unroll
      for (int k = 0; k < 12; ++k) ow[k] = 0;
    }
    uint4* d4 = reinterpret_cast<uint4*>(&dst->w[q][0]);
    d4[0] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    d4[1] = make_uint4(ow[4], ow[5], ow[6], ow[7]);
    d4[2] = make_uint4(ow[8], ow[9], ow[10], ow[11]);
    if (q == 0u) dst->inf = acc.inf;
  }
  if (S == 1u && tid == 0) atomicAdd(&a.counters[gctr + 2], s_off[256]);
  __threadfence_system();
  __syncthreads();
  if (tid == 0) {
    a.counters[wslot] = 0;                                     // this window's ticket word is free for the next call
    const uint32_t done = atomicAdd(&a.counters[gctr], 1u);
    if (done == gctr - 1u) {                                   // the last window of the last MSM: status words, then the flag the host polls
      uint32_t* st = reinterpret_cast<uint32_t*>(a.out_host + (size_t)gctr * nitems);
      st[0] = atomicAdd(&a.counters[gctr + 1], 0u);
      st[1] = atomicAdd(&a.counters[gctr + 2], 0u);
      st[2] = 0; st[3] = 0;
      a.counters[gctr] = 0; a.counters[gctr + 1] = 0; a.counters[gctr + 2] = 0;
      __threadfence_system();
      __hip_atomic_store(a.flag_host, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
