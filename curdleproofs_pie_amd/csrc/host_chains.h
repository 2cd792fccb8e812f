// Host side of the MSM engine, part 2: the window planner and the launch chains (regime A, the single-launch small MSM, regime B).
// Part of the single translation unit csrc/msm_gpu.hip (included there, in this order; not a stand-alone header).
#pragma once

namespace cg1 {

// c > 0: uniform windows of width c (nwin = 255 / c + 1).  c < 0: a BALANCED plan with cmax = -c: the 256 bit positions
// are cut into nwin = ceil(256 / cmax) windows of width cmax (the low ones) or cmax - 1, so the top window keeps
// >= cmax - 2 scalar bits and the recoding carry never leaves it (scalars are < 2^255).
// glv: the same over the 128 bit positions of the halves of the endomorphism split (glv.h: magnitudes < 0.68 * 2^127).
static WinPlan make_plan(int c, bool glv = false) {
  WinPlan pl;
  const int bits = glv ? 127 : 255;
  pl.glv = glv ? 1 : 0;
  if (c > 0) { pl.cmax = c; pl.nwin = bits / c + 1; pl.n_hi = pl.nwin; }
  else {
    const int cm = -c, nw = (bits + 1 + cm - 1) / cm;
    pl.cmax = cm; pl.nwin = nw; pl.n_hi = bits + 1 - nw * (cm - 1);
    if (pl.n_hi < 0) pl.n_hi = 0;        // nw windows of width cm - 1 already cover every position (128 positions at cm = 14: 10 x 13): all narrow
  }
  return pl;
}

// window_c = 0: the plan per input size (tools/gpu_window_sweep.py on MI355X).  The 2-D bucket reduction costs two EC additions
// per BUCKET, the accumulation one per (term, window): mid-size inputs want fewer, fuller buckets than c = 16 gives them.
static int pick_plan_c(size_t n, int auto_plan) {
  const int c = pick_window(n);
  if (!auto_plan || c != 16) return c;
  if (n <= (1u << 14)) return -12;
  if (n <= (3u << 15)) return -13;
  if (n <= (3u << 16)) return -15;
  return 16;
}

int pick_window(size_t n) {
  // Only widths whose TOP window still holds >= min(c-1, 7) scalar bits (255 = (nwin-1)*c + t): with t = 2..3 all
  // n terms of that window fall into <= 8 buckets.  Thresholds from tools/gpu_window_sweep.py on MI355X.
  if (n <= 128) return 4;        // t = 3
  if (n <= 8192) return 8;       // t = 7
  return 16;                     // t = 15
}

// Wait for the context's compute stream.  blocking_sync: sleep until the GPU signals (an event created with
// hipEventBlockingSync) and leave the core to the front-end threads; default: the runtime's spinning wait (lowest latency).
static int wait_stream(Ctx* ctx) {
  if (ctx->blocking_sync) {
    HIPCHK(hipEventRecord(ctx->sync_ev, ctx->stream));
    HIPCHK(hipEventSynchronize(ctx->sync_ev));
    return CG1_OK;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return CG1_OK;
}

// Where an MSM's points come from (all device memory):
//   AFFINE96  n x 96 B standard-form affine records (the C ABI's "affine96")                      -> k_prepare_points
//   BLOBS     n x 144 B host point blobs as G1Point objects hold them (Jacobian, radix 2^384)     -> k_prepare_blobs
//   PREPARED  n x 128 B records + n identity flags made earlier by one of the two (a cg1_vec)    -> nothing to do
struct PtSrc {
  enum Kind { AFFINE96 = 0, BLOBS = 1, PREPARED = 2 } kind = AFFINE96;
  const void* p = nullptr;
  const uint8_t* flags = nullptr;       // PREPARED only
  bool normalised = false;              // BLOBS only: every Z is 0 or 1 (no inversion needed)
  PtSrc() {}
  PtSrc(const void* affine96) : p(affine96) {}
};

// points per lane of k_prepare_blobs<true>: one Fermat inversion per lane, so few lanes for big inputs -- but never fewer than
// ~2 waves per SIMD's worth, where the launch turns from latency- into throughput-bound
static uint32_t blob_points_per_lane(size_t n) {
  size_t K = (n + (1u << 17) - 1) >> 17;
  return (uint32_t)(K < 1 ? 1 : (K > 16 ? 16 : K));
}

// records + flags from `src` into (out, flags_out) on `st`; clears the call's status words (like k_prepare_points)
static void launch_prepare(hipStream_t st, const PtSrc& src, PreparedPoint* out, uint8_t* flags_out, uint32_t n32, uint32_t* status_words) {
  if (src.kind == PtSrc::AFFINE96) {
    hipLaunchKernelGGL(k_prepare_points, dim3((n32 + 255) / 256), dim3(256), 0, st, (const uint32_t*)src.p, out, flags_out, n32, status_words);
  } else if (src.normalised) {
    hipLaunchKernelGGL((k_prepare_blobs<false>), dim3((n32 + 127) / 128), dim3(128), 0, st, (const uint32_t*)src.p, out, flags_out, n32, 1u, status_words);
  } else {
    const uint32_t K = blob_points_per_lane(n32);
    const uint32_t lanes = (n32 + K - 1) / K;
    hipLaunchKernelGGL((k_prepare_blobs<true>), dim3((lanes + 127) / 128), dim3(128), 0, st, (const uint32_t*)src.p, out, flags_out, n32, K, status_words);
  }
}

// Enqueue the whole launch chain of this context's share of an MSM (windows w = rank mod world of the plan) up to the D2H of
// the window sums; nothing waits.
// Hooks of the two-chain form of one call (msm_begin_split): `after_prepare` is recorded on the chain's stream once the prepared
// records exist (the other chain reads them); the chain waits for `before_start` before its first launch and for
// `before_accumulate` in front of k_accumulate; `after_accumulate` is recorded behind k_accumulate.
struct ChainHooks {
  hipEvent_t before_start = nullptr, after_prepare = nullptr, before_accumulate = nullptr, after_accumulate = nullptr;
};

static int msm_enqueue(Ctx* ctx, const PtSrc& src, const void* d_scalars32, size_t n, const WinPlan& plan, int rank, int world,
                       const ChainHooks& hooks = ChainHooks()) {
  ctx->pend.active = false;
  HIPCHK(hipSetDevice(ctx->device));
  const int c = plan.cmax, nwin = plan.nwin;
  // an endomorphism-split call (plan.glv) runs over 2n records -- P_i, then phi(P_i) -- and 2n rows of digits; everything behind the digit
  // kernel sees an MSM of `n` = 2 n_real points
  const size_t n_real = n;
  if (plan.glv) n *= 2;
  const int nlw = win_count(nwin, rank, world);                // `world` is a window selector (kernels_prepare_digits.h win_sel): the share w = rank (mod world), or a run of it
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
  // Between 2^17 and 2^19 terms k_accumulate is already bound by throughput, not by the chain of one chunk, and the lane-per-bucket tail
  // runs (more than 2^18 buckets): there a chunk should hold a WHOLE bucket -- mean load m plus eight standard deviations of its Poisson
  // spread -- so that no bucket is cut, k_bucket_fold finds nothing to do and k_rowcol reads one sum per bucket (profiles/r04_chunk_ab.txt:
  // 2^18 terms 1.22 -> 1.11 ms).  Below 2^22 entries the chain still shows: 20 at most (2^17 terms: 0.94 -> 0.91 ms).
  if (ctx->chunk_rule && world == 1) {
    const uint64_t entries = (uint64_t)n * (uint64_t)nlw;
    if (entries >= (1ull << 20) && entries < (1ull << 21) && L0 < 10u) L0 = 10u;      // 2^16 terms: 0.715 -> 0.695 ms (same file)
    if (entries >= (1ull << 21) && entries < (1ull << 24) && nb_total > (size_t)ctx->rowcol_quad_max) {
      const double mload = (double)n / (double)(1u << bb);
      uint32_t want = (uint32_t)(mload + 8.0 * std::sqrt(mload) + 1.0);
      if (entries < (1ull << 22) && want > 20u) want = 20u;
      if (want > L0) L0 = want;
    }
  }
  int rc = ensure(ctx, n, nb_total, nlw, nitems, L0, src.kind != PtSrc::PREPARED);
  if (rc) return rc;
  hipStream_t st = ctx->stream;
  const uint32_t n32 = (uint32_t)n;
  const uint32_t gn = (n32 + 255) / 256;
  const int profile = ctx->profile;
  auto h0 = std::chrono::steady_clock::now();
  const bool resident = src.kind == PtSrc::PREPARED;
  const PreparedPoint* pts = resident ? static_cast<const PreparedPoint*>(src.p) : ctx->d_pts;
  const uint8_t* flags = resident ? src.flags : ctx->d_flags;
  const size_t nout_words = (size_t)nlw * nitems;
  uint32_t* bad_flag = reinterpret_cast<uint32_t*>(ctx->d_out + nout_words);       // [0] set by the digit kernels: a scalar >= 2^255; [1], [2]: counts
  if (hooks.before_start) HIPCHK(hipStreamWaitEvent(st, hooks.before_start, 0));
  if (profile >= 2) HIPCHK(hipEventRecord(ctx->ev[0], st));
  const uint32_t n_real32 = (uint32_t)n_real;
  if (resident) HIPCHK(hipMemsetAsync(bad_flag, 0, 16, st));                       // (an endomorphism-split call: the source already holds both halves)
  else {
    launch_prepare(st, src, ctx->d_pts, ctx->d_flags, n_real32, bad_flag);
    if (plan.glv) hipLaunchKernelGGL(k_phi_records, dim3((n_real32 + 255) / 256), dim3(256), 0, st, ctx->d_pts, ctx->d_flags, n_real32);
  }
  if (hooks.after_prepare) HIPCHK(hipEventRecord(hooks.after_prepare, st));
  if (profile >= 2) HIPCHK(hipEventRecord(ctx->ev[1], st));
  const uint32_t nblk = (uint32_t)((nb_total + SCAN_ITEMS - 1) / SCAN_ITEMS);
  if (ctx->use_partition_sort && n <= PART_MAX_N) {
    // ---- two-level partition sort: no global atomics
    // bins per window = 2^(bb - sub_bits) <= 128 (the partition kernels' LDS tables); a bin is ONE workgroup of k_bin_sort, so mid sizes
    // want many small bins ("sort_sub_bits": the sub-bucket width, 8 at most; A/B in profiles/r04_sort_bins_ab.txt)
    const uint32_t want_sub = ctx->sort_sub_bits ? (uint32_t)ctx->sort_sub_bits : (n <= ((size_t)1 << 16) ? 7u : 8u);   // 0 = by size: measured
    uint32_t sub_bits = bb < want_sub ? bb : want_sub;
    while (bb - sub_bits > 7u) ++sub_bits;
    const uint32_t nbins = 1u << (bb - sub_bits);
    const uint32_t nslices = (n32 + PART_TILE - 1) / PART_TILE;
    const uint32_t nbc = (uint32_t)nlw * nbins * nslices;
    hipLaunchKernelGGL(k_digits, dim3((n_real32 + 255) / 256), dim3(256), 0, st, (const uint32_t*)d_scalars32, flags, ctx->d_digits, n_real32, plan, rank, world, bad_flag);
    hipLaunchKernelGGL(k_part_count, dim3(nslices, nlw), dim3(256), 0, st, ctx->d_digits, ctx->d_blockcnt, n32, nslices, nbins, sub_bits);
    const uint32_t ublk = (nbc + SCAN_ITEMS - 1) / SCAN_ITEMS;
    if (ctx->scan_one && nbc <= USCAN1_MAX) {
      hipLaunchKernelGGL(k_uscan_one, dim3(1), dim3(1024), 0, st, ctx->d_blockcnt, nbc);
    } else {
      hipLaunchKernelGGL(k_uscan1, dim3(ublk), dim3(256), 0, st, ctx->d_blockcnt, ctx->d_ublocktot, nbc);
      hipLaunchKernelGGL(k_uscan2, dim3(1), dim3(256), 0, st, ctx->d_ublocktot, ublk, ctx->d_blockcnt, nbc);
      hipLaunchKernelGGL(k_uscan3, dim3(ublk), dim3(256), 0, st, ctx->d_ublocktot, ctx->d_blockcnt, nbc);
    }
    if (profile >= 2) HIPCHK(hipEventRecord(ctx->ev[2], st));
    hipLaunchKernelGGL(k_part_scatter, dim3(nslices, nlw), dim3(256), 0, st, ctx->d_digits, ctx->d_blockcnt, ctx->d_part, n32, nslices, nbins, sub_bits, ctx->stage_sort);
    const uint32_t nbt = (uint32_t)nlw * nbins;
    // the slice path for bins beyond k_bin_sort's LDS stage: a bin cannot exceed n entries, and up to twice the stage k_bin_sort's
    // direct path is as fast as three more launches even for all-equal scalars (2^14 terms: 0.766 -> 0.752 ms, uniform 0.409 -> 0.401)
    const int big_bins = ctx->big_bins && n32 > 2u * BIN_STAGE;
    hipLaunchKernelGGL(k_slice_plan, dim3(1), dim3(256), 0, st, ctx->d_blockcnt, nbt, nslices, ctx->d_slice_base, ctx->d_bigflag, big_bins);
    hipLaunchKernelGGL(k_bin_sort, dim3(nbt), dim3(256), 0, st, ctx->d_part, ctx->d_blockcnt, ctx->d_hist, ctx->d_sorted, nbt, nslices, sub_bits, ctx->stage_sort, ctx->d_bigflag);
    if (big_bins) {
      const uint32_t max_slices = (uint32_t)(((size_t)n * (size_t)nlw) / SLICE + nbt + 1);
      hipLaunchKernelGGL(k_slice_count, dim3(max_slices), dim3(256), 0, st, ctx->d_part, ctx->d_blockcnt, nbt, nslices, ctx->d_slice_base, ctx->d_slicehist);
      hipLaunchKernelGGL(k_slice_prefix, dim3(nbt), dim3(256), 0, st, ctx->d_blockcnt, nslices, sub_bits, ctx->d_slice_base, ctx->d_bigflag, ctx->d_slicehist, ctx->d_subbase, ctx->d_hist);
      hipLaunchKernelGGL(k_slice_scatter, dim3(max_slices), dim3(256), 0, st, ctx->d_part, ctx->d_blockcnt, nbt, nslices, ctx->d_slice_base, ctx->d_slicehist, ctx->d_subbase, ctx->d_sorted);
    }
    if (profile >= 2) HIPCHK(hipEventRecord(ctx->ev[3], st));
    if (ctx->scan_one && nb_total <= SCAN1_MAX) {
      hipLaunchKernelGGL(k_scan_one, dim3(1), dim3(1024), 0, st, ctx->d_hist, ctx->d_off, ctx->d_choff, (uint32_t)nb_total, L0);
    } else {
      hipLaunchKernelGGL(k_scan1, dim3(nblk), dim3(256), 0, st, ctx->d_hist, ctx->d_off, ctx->d_choff, ctx->d_blocktot, (uint32_t)nb_total, L0);
      hipLaunchKernelGGL(k_scan2, dim3(1), dim3(256), 0, st, ctx->d_blocktot, nblk, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
      hipLaunchKernelGGL(k_scan3, dim3(nblk), dim3(256), 0, st, ctx->d_blocktot, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
    }
  } else {
    // ---- global-atomic counting sort (any n < 2^31)
    HIPCHK(hipMemsetAsync(ctx->d_hist, 0, nb_total * 4, st));
    hipLaunchKernelGGL(k_hist, dim3(gn), dim3(256), 0, st, (const uint32_t*)d_scalars32, flags, ctx->d_hist, n32, plan, rank, world, bad_flag);
    if (profile >= 2) HIPCHK(hipEventRecord(ctx->ev[2], st));
    hipLaunchKernelGGL(k_scan1, dim3(nblk), dim3(256), 0, st, ctx->d_hist, ctx->d_off, ctx->d_choff, ctx->d_blocktot, (uint32_t)nb_total, L0);
    hipLaunchKernelGGL(k_scan2, dim3(1), dim3(256), 0, st, ctx->d_blocktot, nblk, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
    hipLaunchKernelGGL(k_scan3, dim3(nblk), dim3(256), 0, st, ctx->d_blocktot, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
    if (profile >= 2) HIPCHK(hipEventRecord(ctx->ev[3], st));
    hipLaunchKernelGGL(k_scatter, dim3(gn), dim3(256), 0, st, (const uint32_t*)d_scalars32, flags, ctx->d_hist, ctx->d_off, ctx->d_sorted, n32, plan, rank, world);
  }
  // one memset: chunk-length histogram, the any_multi flag, combined[] and the heavy-bucket count (ensure() laid them out together)
  HIPCHK(hipMemsetAsync(ctx->d_zblock, 0, zblock_clear_bytes(ctx), st));
  hipLaunchKernelGGL(k_chunk_desc, dim3((uint32_t)std::min<size_t>(CHUNK_DESC_BLOCKS, (nb_total + 255) / 256)), dim3(256), 0, st, ctx->d_off, ctx->d_choff, ctx->d_desc, ctx->d_lenhist, ctx->d_heavy, (uint32_t)ctx->cap_heavy, (uint32_t)nb_total, L0, ctx->d_any_multi);
  const size_t max_chunks = nb_total + (n * (size_t)nlw) / L0 + 1;
  const uint32_t gchunks = (uint32_t)((max_chunks + 255) / 256);
  hipLaunchKernelGGL(k_len_scan, dim3(1), dim3(256), 0, st, ctx->d_lenhist, ctx->d_lenhist + LEN_BINS, ctx->d_off + nb_total, ctx->d_choff + nb_total, bad_flag);
  hipLaunchKernelGGL(k_order, dim3((gchunks + ORDER_PER - 1) / ORDER_PER), dim3(256), 0, st, ctx->d_desc, ctx->d_choff + nb_total, ctx->d_lenhist + LEN_BINS, ctx->d_order);
  if (hooks.before_accumulate) HIPCHK(hipStreamWaitEvent(st, hooks.before_accumulate, 0));
  if (profile >= 1) HIPCHK(hipEventRecord(ctx->ev[4], st));
  hipLaunchKernelGGL(k_accumulate, dim3(gchunks), dim3(256), 0, st, ctx->d_desc, ctx->d_choff + nb_total, ctx->d_order, ctx->d_sorted, pts, ctx->d_sums);
  if (profile >= 1) HIPCHK(hipEventRecord(ctx->ev[5], st));
  if (hooks.after_accumulate) HIPCHK(hipEventRecord(hooks.after_accumulate, st));
  hipLaunchKernelGGL(k_heavy_combine, dim3(512), dim3(256), 0, st, ctx->d_heavy, (uint32_t)ctx->cap_heavy, ctx->d_choff, ctx->d_sums, ctx->d_combined);
  // k_rowcol_quad (every addition by a DPP quad) only where the reduction is a pure latency chain: a few thousand buckets
  const bool small_quad = ctx->quad && ctx->rowcol_quad && nb_total <= (size_t)ctx->rowcol_quad_max;
  // Buckets cut into 2..16 chunks are folded into their first slot before the row / column sums (k_rowcol_quad requires it;
  // k_rowcol / k_seg_reduce could add the chunk sums themselves -- bucket_sum -- but the divergent trip counts inside their lanes
  // cost more than the separate pass: profiles/r03_rowcol_ab.txt).
  // k_rowcol_quad_row can add a bucket's chunk sums itself: worth it only for a few thousand buckets (2^12 terms 0.354 -> 0.336 ms; at 2^15 -
  // 2^16 the uneven trip counts of a wave's quads cost more than the separate pass: 0.605 -> 0.657, profiles/r05_tree_row_ab.txt)
  const bool rows_fold = small_quad && use2d && ctx->rowcol_row && ctx->tree_row && nb_total <= 8192;
  if (rows_fold) {}
  else if (small_quad && ctx->fold_quad)
    hipLaunchKernelGGL(k_bucket_fold_quad, dim3((uint32_t)((nb_total * 4 + 255) / 256)), dim3(256), 0, st, ctx->d_choff, ctx->d_sums, ctx->d_combined, (uint32_t)nb_total, ctx->d_any_multi);
  else if (ctx->fold_pass)
    hipLaunchKernelGGL(k_bucket_fold, dim3((uint32_t)((nb_total + 255) / 256)), dim3(256), 0, st, ctx->d_choff, ctx->d_sums, ctx->d_combined, (uint32_t)nb_total, ctx->d_any_multi);
  bool tree_exports = false;
  if (use2d) {
    const uint32_t R = 1u << hb2, Cn = 1u << lb2;
    const uint32_t lpr = Cn < 32u ? Cn : 32u, lpc = R < 16u ? R : 16u;
    const uint32_t nrow_blocks = ((uint32_t)nlw * R + (256u / lpr) - 1u) / (256u / lpr);
    const uint32_t ncol_blocks = ((uint32_t)nlw * Cn + (256u / lpc) - 1u) / (256u / lpc);
    PointSum* rowsum = ctx->d_segrun;                     // reuse the segment buffers (>= nb_total records each)
    PointSum* colsum = ctx->d_segtot;
    if (small_quad) {
      // quads per row / column: 16, 8 or 4 (a wave carries 1, 2 or 4 rows).  Measured (profiles/r04_rowcol_ab.txt, fold + row / column
      // sums at 2^12 .. 2^16 terms): a serial element costs a quad ~8 us, a shuffle level ~23 us (56 words through ds_bpermute), and
      // 2 560 one-wave rows are 1.25 rounds of the 2 048 resident waves -- so FEW quads per row win: 4 where rows and columns are equally
      // long (119 / 104 / 97 us at 2^12, 248 / 229 / 216 at 2^16 for 16 / 8 / 4 quads), 8 where they are not (2^14: 158 / 150 / 167).
      uint32_t lgq = (R == Cn) ? 2u : 3u;
      if (ctx->rowcol_lgq >= 2 && ctx->rowcol_lgq <= 4) lgq = (uint32_t)ctx->rowcol_lgq;
      const uint32_t rc_waves = ((uint32_t)nlw * (R + Cn) + (16u >> lgq) - 1u) / (16u >> lgq);
      if (ctx->rowcol_row && ctx->tree_row)              // one wave per row / column, the cross-quad levels on rows (its sums are in row form: k_small_tree_row reads them)
        hipLaunchKernelGGL(k_rowcol_quad_row, dim3(((uint32_t)nlw * (R + Cn) + 3u) / 4u), dim3(256), 0, st, ctx->d_choff, ctx->d_sums, ctx->d_combined,
                           rowsum, colsum, (uint32_t)nlw, hb2, lb2);
      else
      hipLaunchKernelGGL(k_rowcol_quad, dim3((rc_waves + 3u) / 4u), dim3(256), 0, st, ctx->d_choff, ctx->d_sums,
                         rowsum, colsum, (uint32_t)nlw, hb2, lb2, lgq);
    }
    else
      hipLaunchKernelGGL(k_rowcol, dim3(nrow_blocks + ncol_blocks), dim3(256), 0, st, ctx->d_choff, ctx->d_sums, ctx->d_combined,
                         rowsum, colsum, (uint32_t)nlw, hb2, lb2, nrow_blocks, ctx->quad);
    if (profile >= 2) HIPCHK(hipEventRecord(ctx->ev[6], st));
    // 4 lanes per element of the longer of the two sums (2^hb rows, 2^lb columns), at most 512 threads: no idle quads in the block
    const uint32_t tree_threads = std::min<uint32_t>(512u, std::max<uint32_t>(64u, 4u << std::max(hb2, lb2)) >> (ctx->tree_shift >= 0 ? ctx->tree_shift : (ctx->tree_half ? 1 : 0)));
    if (ctx->tree_row) {
      uint32_t W = (1u << std::max(hb2, lb2)) >> 3;              // ~8 selected elements per wave at most, 16 waves at most
      W = W < 1u ? 1u : (W > 16u ? 16u : W);
      // zero-copy calls: the items, the status words and the flag word go to the mapped host records from this kernel (no k_export_host)
      tree_exports = ctx->zero_copy != 0;
      if (tree_exports) ++ctx->seq;
      hipLaunchKernelGGL(k_small_tree_row, dim3(nitems, nlw), dim3(64u * W), 0, st, rowsum, colsum, ctx->d_out, hb2, lb2,
                         tree_exports ? ctx->h_out_dev : (PointWords*)nullptr, bad_flag, ctx->h_flag_dev, ctx->seq);
    }
    else if (ctx->quad) hipLaunchKernelGGL(k_small_tree_quad, dim3(nitems, nlw), dim3(std::max<uint32_t>(64u, tree_threads)), 0, st, rowsum, colsum, ctx->d_out, hb2, lb2);
    else hipLaunchKernelGGL(k_small_tree, dim3(nitems, nlw), dim3(256), 0, st, rowsum, colsum, ctx->d_out, hb2, lb2);
  } else {
    const uint32_t nseg_total = (uint32_t)(nb_total / m);
    hipLaunchKernelGGL(k_seg_reduce, dim3((nseg_total + 255) / 256), dim3(256), 0, st, ctx->d_choff, ctx->d_sums, ctx->d_combined, ctx->d_segrun, ctx->d_segtot, nseg_total, m);
    if (profile >= 2) HIPCHK(hipEventRecord(ctx->ev[6], st));
    uint32_t S = (J + BT_ELEMS - 1) / BT_ELEMS; if (S < 1) S = 1; if (S > 64) S = 64;   // J <= 2^15 / seg_m
    hipLaunchKernelGGL(k_bit_tree, dim3(nitems, nlw, S), dim3(256), 0, st, ctx->d_segrun, ctx->d_segtot, ctx->d_partial, J);
    hipLaunchKernelGGL(k_bit_tree_final, dim3((uint32_t)(nitems * nlw)), dim3(64), 0, st, ctx->d_partial, ctx->d_out, S);
  }
  const bool zc = ctx->zero_copy != 0;
  if (zc && tree_exports) {
    // (k_small_tree_row has written the records, the status words and the flag)
  } else if (zc) {
    // the window sums + status words go straight into mapped host memory, then the call's sequence number into the flag word the
    // host polls: no DMA copy to set up, no stream wait to wake from (~25 us per call, all of it on the critical path of a small MSM)
    ++ctx->seq;
    const uint32_t nvec = (uint32_t)((((size_t)nlw * nitems + 1) * sizeof(PointWords)) / 16);
    hipLaunchKernelGGL(k_export_host, dim3(1), dim3(1024), 0, st, reinterpret_cast<const uint4*>(ctx->d_out), reinterpret_cast<uint4*>(ctx->h_out_dev), nvec,
                       ctx->h_flag_dev, ctx->seq);
  } else {
    HIPCHK(hipMemcpyAsync(ctx->h_out, ctx->d_out, ((size_t)nlw * nitems + 1) * sizeof(PointWords), hipMemcpyDeviceToHost, st));
  }
  if (profile >= 2) HIPCHK(hipEventRecord(ctx->ev[7], st));
  auto h1 = std::chrono::steady_clock::now();
  Ctx::Pending& pd = ctx->pend;
  pd.zero_copy = zc; pd.seq = ctx->seq; pd.hout = ctx->h_out;
  pd.active = true; pd.c = c; pd.plan = plan; pd.rank = rank; pd.world = world; pd.nlw = nlw; pd.nbits = nbits; pd.m = m; pd.lb2 = lb2; pd.hb2 = hb2;
  pd.nitems = nitems; pd.use2d = use2d; pd.profile = profile; pd.nout_words = nout_words; pd.h0 = h0; pd.h1 = h1;
  pd.arm_helpers = ctx->arm_helpers && ctx->host_split && ctx->horner_threads > 1 && n <= ((size_t)1 << 18);
  return CG1_OK;
}

// Wait for what msm_enqueue queued on this context, then the host Horner tail over its windows.
static int msm_finish(Ctx* ctx, cg1h::jac& result) {
  result = cg1h::jac_identity();
  if (!ctx->pend.active) return CG1_OK;
  const Ctx::Pending pd = ctx->pend;
  ctx->pend.active = false;
  const int c = pd.c, rank = pd.rank, world = pd.world, nlw = pd.nlw, nbits = pd.nbits;
  (void)c;
  const uint32_t m = pd.m, lb2 = pd.lb2, hb2 = pd.hb2, nitems = pd.nitems;
  const bool use2d = pd.use2d;
  const size_t nout_words = pd.nout_words;
  const auto h0 = pd.h0, h1 = pd.h1;
  HIPCHK(hipSetDevice(ctx->device));
  // helpers armed below spin for a job for up to 2 ms: on every way out of this function (a failed stream, a bad scalar, a tail that needs
  // fewer threads) the ones still spinning are told to stand down
  struct StandDown { Ctx* c; bool on; ~StandDown() { if (on) for (int j = 0; j < 3; ++j) c->helper[j].disarm(); } } stand_down{ctx, false};
  if (pd.zero_copy && !ctx->blocking_sync && pd.profile < 2) {
    if (pd.arm_helpers) for (int j = 0; j < 3 && j + 1 < ctx->horner_threads; ++j) ctx->helper[j].arm();
    stand_down.on = pd.arm_helpers;
    // poll the flag word k_export_host writes last; look at the stream now and then so that a failed launch cannot hang us
    volatile uint32_t* flag = ctx->h_flag;
    for (uint32_t spins = 0; *flag != pd.seq; ++spins) {
      if ((spins & 0x3fffu) == 0x3fffu) {
        hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipSuccess) { if (*flag != pd.seq) { snprintf(ctx->err, sizeof ctx->err, "the stream drained without the export flag"); return CG1_ERR_HIP; } break; }
        if (q != hipErrorNotReady) { snprintf(ctx->err, sizeof ctx->err, "stream failed: %s", hipGetErrorString(q)); return CG1_ERR_HIP; }
      }
      __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else {
    int wrc = wait_stream(ctx); if (wrc) return wrc;
  }
  HIPCHK(hipGetLastError());
  {
    const uint32_t* st_words = reinterpret_cast<const uint32_t*>(pd.hout + nout_words);    // [0] bad scalar, [1] entries, [2] chunks
    ctx->last_entries = st_words[1]; ctx->last_chunks = st_words[2];
    if (st_words[0]) {
      snprintf(ctx->err, sizeof ctx->err, "a scalar is >= 2^255: scalar32 must be a canonical Fr element (< r)");
      return CG1_ERR_ENCODING;
    }
  }
  auto h2 = std::chrono::steady_clock::now();
  { int erc = read_phase_events(ctx, pd.profile); if (erc) return erc; }
  ctx->last_c = c;

  // ---- host tail: ONE Horner over global bit positions.
  //   result = sum over the exported points P of 2^e(P) P, with (window w of the plan starts at bit off[w]):
  //   2-D reduction:  e(T0_w) = off[w];  e(column bit k) = off[w] + k (k < lb);  e(row bit k) = off[w] + lb + k (k < hb)
  //   1-D fallback:   e(T_w) = off[w];   e(Y_{w,b}) = off[w] + log2(m) + b
  // (a window narrower than cmax leaves its top row bits empty: their points are the identity and are skipped)
  auto t0 = std::chrono::steady_clock::now();
  ctx->host_ms[0] = std::chrono::duration<float, std::milli>(h1 - h0).count();
  ctx->host_ms[1] = std::chrono::duration<float, std::milli>(h2 - h1).count();
  ctx->host_ms[2] = std::chrono::duration<float, std::milli>(t0 - h2).count();
  int lm = 0; while ((1u << lm) < m) ++lm;
  const WinPlan& plan = pd.plan;
  constexpr int EMAX = 2 * 256 + 64;
  std::vector<std::pair<int, const PointWords*>> items;          // (exponent, point), then grouped by exponent
  items.reserve((size_t)nlw * nitems);
  int e_top = 0;
  for (int lw = 0; lw < nlw; ++lw) {
    const int w = win_global(lw, rank, world), base = plan.off(w);
    const PointWords* row = pd.hout + (size_t)lw * nitems;
    auto put = [&](int e, const PointWords* p) { if (!p->inf && e >= 0 && e < EMAX) { items.emplace_back(e, p); if (e > e_top) e_top = e; } };      // (a plan never leaves [0, EMAX): the guard keeps a planner bug off the stack)
    put(base, &row[0]);
    if (use2d) {
      for (uint32_t k = 0; k < hb2; ++k) put(base + (int)lb2 + (int)k, &row[1 + k]);
      for (uint32_t k = 0; k < lb2; ++k) put(base + (int)k, &row[1 + hb2 + k]);
    } else {
      for (int b2 = 0; b2 < nbits; ++b2) put(base + lm + b2, &row[1 + b2]);
    }
  }
  uint16_t first[EMAX + 1];                                       // counting sort by exponent
  memset(first, 0, sizeof first);
  for (const auto& it : items) ++first[it.first + 1];
  for (int e = 0; e < EMAX; ++e) first[e + 1] = (uint16_t)(first[e + 1] + first[e]);
  std::vector<const PointWords*> byexp(items.size());
  {
    uint16_t cur[EMAX];
    memcpy(cur, first, sizeof cur);
    for (const auto& it : items) byexp[cur[it.first]++] = it.second;
  }
  // horner(lo, hi) = sum_{e in [lo, hi]} 2^(e - lo) * (points of weight 2^e)
  auto horner = [&](int lo, int hi) {
    cg1h::jac a = cg1h::jac_identity();
    for (int e = hi; e >= lo; --e) {
      a = cg1h::jac_dbl(a);
      for (uint16_t k = first[e]; k < first[e + 1]; ++k) a = cg1h::jac_add(a, jac_from_words(*byexp[k]));
    }
    return a;
  };
  cg1h::jac acc;
  const int nth = (!ctx->host_split || e_top < 96) ? 1 : (ctx->horner_threads >= 4 && e_top >= 112 ? 4 : 2);      // (112: the 128-position plans of the endomorphism split)
  if (nth > 1) {
    // the exponent range cut into nth parts: part j (on its own thread) forms horner(lo_j, hi_j) and then doubles it lo_j times, so
    // every part ends with its full weight and the parts are simply added.  The critical path is the top part: e_top doublings,
    // but only a fraction of the additions.
    // The cut is NOT even: part j costs (lo_{j+1}) doublings + its own additions, so the top part gets the narrowest range.  With a
    // doubling at 7 and an addition at 16 field-multiplication times the largest part cost C is found by bisection (parts filled from
    // the bottom up to C each): four threads end ~18 % sooner than with equal ranges (255 doublings + ~40 additions on the top part).
    cg1h::jac part[4];
    int lo[5];
    {
      constexpr long DBL = 7, ADD = 16;
      auto fill = [&](long C, int* cut) {                          // greedy cut for a part-cost limit C; true if nth parts suffice
        int e = 0;
        for (int j = 0; j < nth; ++j) {
          cut[j] = e;
          long adds = 0;
          while (e <= e_top && DBL * (e + 1) + ADD * (adds + (first[e + 1] - first[e])) <= C) { adds += first[e + 1] - first[e]; ++e; }
        }
        cut[nth] = e_top + 1;
        return e > e_top;
      };
      long lo_c = DBL * (e_top + 1), hi_c = DBL * (e_top + 1) + ADD * (long)items.size();
      int cut[5];
      while (lo_c < hi_c) {
        const long mid = (lo_c + hi_c) / 2;
        if (fill(mid, cut)) hi_c = mid; else lo_c = mid + 1;
      }
      fill(hi_c, cut);
      for (int j = 0; j <= nth; ++j) lo[j] = cut[j];
    }
    auto run_part = [&](int j) {
      cg1h::jac a = horner(lo[j], lo[j + 1] - 1);
      for (int k = 0; k < lo[j]; ++k) a = cg1h::jac_dbl(a);
      part[j] = a;
    };
    for (int j = 0; j + 1 < nth; ++j) ctx->helper[j].run([&, j]() { run_part(j); });
    if (pd.arm_helpers) for (int j = nth - 1 < 0 ? 0 : nth - 1; j < 3; ++j) ctx->helper[j].disarm();      // armed for a wider tail than this call has
    run_part(nth - 1);
    acc = part[nth - 1];
    for (int j = 0; j + 1 < nth; ++j) { ctx->helper[j].wait(); acc = cg1h::jac_add(acc, part[j]); }
  } else {
    if (pd.arm_helpers) for (int j = 0; j < 3; ++j) ctx->helper[j].disarm();
    acc = horner(0, e_top);
  }
  result = acc;
  ctx->host_tail_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ctx->host_ms[3] = ctx->host_tail_ms;
  return CG1_OK;
}

// The window width k_msm_small runs a call of n terms with (uniform signed windows; 2^(c-1) <= 256 buckets fit one workgroup's
// LDS sort; 5 is left out: its top window would hold nothing but the recoding carry).  Chosen so that a slice of <= 256 terms puts
// a handful of entries into a bucket: every EC addition of the kernel is a ~10 us step of a dependent chain, and the reduction costs
// ~log2(buckets) + 4 of them per window whatever n is, so few buckets (64 at c = 7) beat the wider windows the entry count alone
// would suggest (measured, profiles/r04_small_msm.txt: n = 627 at c = 9 waits 232 us for the GPU, at c = 7 ...).
// (windows x slices must stay within ONE round of workgroups for a single MSM -- SM_ONE_ROUND = the chip's 256 CUs, a workgroup of
// k_msm_small fills one: 1 391 terms at c = 7 are 222 workgroups and take 0.35 ms, 2 048 are 296 = two rounds and take 0.47, more than
// the launch chain's 0.40 (profiles/r04_small_msm.txt) -- so from 1 537 terms on the plan is c = 8: 32 windows x 8 slices = 256)
static int pick_small_c(size_t n) {
  if (n <= 24) return 4;
  if (n <= 96) return 6;
  if (n <= 6 * SM_SLICE) return 7;
  return 8;
}

// One launch (two when un-normalised blobs have to be inverted first) for an MSM of n <= SM_MAX_N terms -- or for M <= SM_MAX_MSMS
// independent ones of at most max_n terms each (d_offs: their M + 1 term offsets on the device); fills ctx->pend like msm_enqueue, so
// msm_finish polls the same flag and runs the same host Horner (M = 1), or msm_small_batched_finish does (M > 1).
static int msm_enqueue_small(Ctx* ctx, const PtSrc& src, const void* d_scalars32, size_t n, int c, uint32_t M = 1, const uint32_t* d_offs = nullptr, size_t max_n = 0,
                             bool glv = false) {
  ctx->pend.active = false;
  HIPCHK(hipSetDevice(ctx->device));
  const WinPlan plan = make_plan(c, glv);
  const uint32_t nwin = (uint32_t)plan.nwin, bb = (uint32_t)c - 1u, lb2 = (bb + 1u) / 2u, hb2 = bb - lb2, nitems = 1u + hb2 + lb2;
  if (M == 1) max_n = n;
  const uint32_t S = (uint32_t)(((glv ? 2 * max_n : max_n) + SM_SLICE - 1) / SM_SLICE);      // (split: slices of the 2n entries)
  auto h0 = std::chrono::steady_clock::now();
  if (!ctx->h_small_out) {
    HIPCHK(hipHostMalloc((void**)&ctx->h_small_out, ((size_t)SM_MAX_MSMS * 64 * 9 + 1) * sizeof(PointWords), hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void**)&ctx->h_small_out_dev, ctx->h_small_out, 0));
    HIPCHK(hipMalloc(&ctx->d_small_ctr, (SM_MAX_MSMS * 64 + 8) * 4));
    // on the context's own stream: it is a non-blocking stream, which a memset on the null stream would NOT be ordered with -- the
    // first launch could find its tickets zeroed under its feet ("the stream drained without the export flag")
    HIPCHK(hipMemsetAsync(ctx->d_small_ctr, 0, (SM_MAX_MSMS * 64 + 8) * 4, ctx->stream));
  }
  const size_t need_partial = (size_t)M * nwin * S * nitems;
  if (S > 1 && need_partial > ctx->cap_small_partial) {
    if (ctx->d_small_partial) (void)hipFree(ctx->d_small_partial);
    ctx->d_small_partial = nullptr; ctx->cap_small_partial = 0;
    HIPCHK(hipMalloc(&ctx->d_small_partial, need_partial * sizeof(PointSum)));
    ctx->cap_small_partial = need_partial;
  }
  hipStream_t st = ctx->stream;
  SmallArgs a;
  a.src = src.p; a.flags = src.flags; a.scalars = static_cast<const uint32_t*>(d_scalars32); a.offs = d_offs;
  a.n = (uint32_t)n; a.M = M; a.S = S; a.c = (uint32_t)c; a.nwin = nwin; a.hb = hb2; a.lb = lb2; a.nitems = nitems;
  a.partial = ctx->d_small_partial; a.counters = ctx->d_small_ctr;
  a.out_host = ctx->h_small_out_dev; a.flag_host = ctx->h_flag_dev; a.seq = ++ctx->seq;
  a.row_tail = ctx->small_row_tail ? 1u : 0u;
  a.glv = glv ? 1u : 0u;
  int kind = (int)src.kind;
  if (src.kind == PtSrc::BLOBS && !src.normalised) {           // invert first (one lane per point), then run on the prepared records
    if (n > ctx->cap_small_pts) {
      if (ctx->d_small_pts) (void)hipFree(ctx->d_small_pts);
      if (ctx->d_small_flags) (void)hipFree(ctx->d_small_flags);
      ctx->d_small_pts = nullptr; ctx->d_small_flags = nullptr; ctx->cap_small_pts = 0;
      const size_t cap = n < SM_MAX_N ? SM_MAX_N : n;
      HIPCHK(hipMalloc(&ctx->d_small_pts, cap * sizeof(PreparedPoint)));
      HIPCHK(hipMalloc(&ctx->d_small_flags, cap + 16));
      ctx->cap_small_pts = cap;
    }
    launch_prepare(st, src, ctx->d_small_pts, ctx->d_small_flags, (uint32_t)n, nullptr);
    a.src = ctx->d_small_pts; a.flags = ctx->d_small_flags;
    kind = (int)PtSrc::PREPARED;
  }
  const dim3 grid(nwin, S, M), block(512);
  if (kind == (int)PtSrc::AFFINE96) hipLaunchKernelGGL((k_msm_small<0>), grid, block, 0, st, a);
  else if (kind == (int)PtSrc::BLOBS) hipLaunchKernelGGL((k_msm_small<1>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((k_msm_small<2>), grid, block, 0, st, a);
  auto h1 = std::chrono::steady_clock::now();
  Ctx::Pending& pd = ctx->pend;
  pd.zero_copy = true; pd.seq = ctx->seq; pd.hout = ctx->h_small_out;
  pd.active = true; pd.c = c; pd.plan = plan; pd.rank = 0; pd.world = 1; pd.nlw = (int)nwin; pd.nbits = 0; pd.m = 1; pd.lb2 = lb2; pd.hb2 = hb2;
  pd.nitems = nitems; pd.use2d = true; pd.profile = 0; pd.nout_words = (size_t)M * nwin * nitems; pd.h0 = h0; pd.h1 = h1;
  pd.arm_helpers = ctx->arm_helpers && ctx->host_split && ctx->horner_threads > 1 && M == 1;
  return CG1_OK;
}

// sum over a window-major block of exported 2-D items (1 + hb + lb per window, uniform plan) of 2^e(P) P: one Horner from the top bit
static cg1h::jac horner_2d_items(const PointWords* rows, const WinPlan& plan, uint32_t nitems, uint32_t hb2, uint32_t lb2) {
  constexpr int EMAX = 2 * 256 + 64;
  std::vector<std::pair<int, const PointWords*>> items;
  items.reserve((size_t)plan.nwin * nitems);
  int e_top = 0;
  for (int w = 0; w < plan.nwin; ++w) {
    const int base = plan.off(w);
    const PointWords* row = rows + (size_t)w * nitems;
    auto put = [&](int e, const PointWords* p) { if (!p->inf) { items.emplace_back(e, p); if (e > e_top) e_top = e; } };
    put(base, &row[0]);
    for (uint32_t k = 0; k < hb2; ++k) put(base + (int)lb2 + (int)k, &row[1 + k]);
    for (uint32_t k = 0; k < lb2; ++k) put(base + (int)k, &row[1 + hb2 + k]);
  }
  uint16_t first[EMAX + 1];
  memset(first, 0, sizeof first);
  for (const auto& it : items) ++first[it.first + 1];
  for (int e = 0; e < EMAX; ++e) first[e + 1] = (uint16_t)(first[e + 1] + first[e]);
  std::vector<const PointWords*> byexp(items.size());
  {
    uint16_t cur[EMAX];
    memcpy(cur, first, sizeof cur);
    for (const auto& it : items) byexp[cur[it.first]++] = it.second;
  }
  cg1h::jac a = cg1h::jac_identity();
  for (int e = e_top; e >= 0; --e) {
    a = cg1h::jac_dbl(a);
    for (uint16_t k = first[e]; k < first[e + 1]; ++k) a = cg1h::jac_add(a, jac_from_words(*byexp[k]));
  }
  return a;
}

// Wait for a launch of M > 1 small MSMs and run their M host Horners (one thread each, up to four at a time).
static int msm_small_batched_finish(Ctx* ctx, uint32_t M, std::vector<cg1h::jac>& results) {
  const Ctx::Pending pd = ctx->pend;
  ctx->pend.active = false;
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->blocking_sync) {
    volatile uint32_t* flag = ctx->h_flag;
    for (uint32_t spins = 0; *flag != pd.seq; ++spins) {
      if ((spins & 0x3fffu) == 0x3fffu) {
        hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipSuccess) { if (*flag != pd.seq) { snprintf(ctx->err, sizeof ctx->err, "the stream drained without the export flag"); return CG1_ERR_HIP; } break; }
        if (q != hipErrorNotReady) { snprintf(ctx->err, sizeof ctx->err, "stream failed: %s", hipGetErrorString(q)); return CG1_ERR_HIP; }
      }
      __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  } else {
    int wrc = wait_stream(ctx); if (wrc) return wrc;
  }
  HIPCHK(hipGetLastError());
  const uint32_t* st_words = reinterpret_cast<const uint32_t*>(pd.hout + pd.nout_words);
  ctx->last_entries = st_words[1]; ctx->last_chunks = 0;
  if (st_words[0]) {
    snprintf(ctx->err, sizeof ctx->err, "a scalar is >= 2^255: scalar32 must be a canonical Fr element (< r)");
    return CG1_ERR_ENCODING;
  }
  auto t0 = std::chrono::steady_clock::now();
  const size_t per = (size_t)pd.plan.nwin * pd.nitems;
  auto one = [&](size_t j) { results[j] = horner_2d_items(pd.hout + j * per, pd.plan, pd.nitems, pd.hb2, pd.lb2); };
  if (M <= 4) {
    const size_t nth = std::min<size_t>(4, M);
    for (size_t t = 1; t < nth; ++t) ctx->helper[t - 1].run([&, t]() { for (size_t j = t; j < M; j += nth) one(j); });
    for (size_t j = 0; j < M; j += nth) one(j);
    for (size_t t = 1; t < nth; ++t) ctx->helper[t - 1].wait();
  } else {                                             // more Horners than the context's own helpers: the process's worker pool, one Horner at a time per thread
    std::atomic<size_t> next{0};
    std::function<void()> work = [&]() { for (;;) { const size_t j = next.fetch_add(1); if (j >= M) return; one(j); } };
    Pool& pool = Pool::get();
    pool.run(work, std::min<size_t>(M, pool.size() + 1));
  }
  auto t1 = std::chrono::steady_clock::now();
  ctx->host_ms[0] = std::chrono::duration<float, std::milli>(pd.h1 - pd.h0).count();
  ctx->host_ms[1] = std::chrono::duration<float, std::milli>(t0 - pd.h1).count();
  ctx->host_ms[2] = 0;
  ctx->host_ms[3] = ctx->host_tail_ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
  for (int i = 0; i < CG1_NPHASE; ++i) ctx->phase_ms[i] = 0.f;
  ctx->last_c = pd.c;
  ctx->last_acc_launches = 0;
  return CG1_OK;
}

static Ctx* child_of(Ctx* ctx);
static int msm_begin_split(Ctx* ctx, const PtSrc& src, const void* d_scalars32, size_t n, const WinPlan& plan, int rank, int world);

// One MSM: this context's share (windows w = rank mod world) of sum_i scalar_i * point_i.
// c = 0: automatic plan; 4..16: uniform windows of that width; -16..-4: the balanced plan with cmax = -c.
// msm_begin enqueues the whole launch chain and returns; msm_end waits for it and runs the host tail.  Two contexts on one
// GPU can thus keep two MSMs in flight: the sort phases of the next one run under this one's k_accumulate (they need few
// registers and co-reside with its waves), and this one's reduction tree, D2H and host Horner run under the next one's.
int msm_begin(Ctx* ctx, const PtSrc& src, const void* d_scalars32, size_t n, int c, int rank, int world) {
  ctx->pend.active = false;
  ctx->pend_c = 0;
  if (n == 0) return CG1_OK;
  if (n >= (1ull << 31)) { snprintf(ctx->err, sizeof ctx->err, "n too large"); return CG1_ERR_ARG; }
  if (world < 1 || world > 255 || rank < 0 || rank >= world) { snprintf(ctx->err, sizeof ctx->err, "bad window shard %d/%d", rank, world); return CG1_ERR_ARG; }
  // the single-launch kernel; with "glv" (the caller vouches for G1) over the 2n entries of the endomorphism split when they fit
  for (int pass = (ctx->glv && 2 * n <= SM_MAX_N) ? 1 : 0; pass >= 0; --pass) {
    const bool sglv = pass == 1;
    const size_t nn = sglv ? 2 * n : n;
    const int bits = sglv ? 127 : 255;
    if (ctx->small_msm && nn <= SM_MAX_N && world == 1 && (c == 0 || (c >= 4 && c <= 9 && c != 5)) &&
        (size_t)(bits / (c ? c : pick_small_c(nn)) + 1) * ((nn + SM_SLICE - 1) / SM_SLICE) <= SM_ONE_ROUND) {
      if (c == 0) c = pick_small_c(nn);
      ctx->pend_c = c;
      return msm_enqueue_small(ctx, src, d_scalars32, n, c, 1, nullptr, 0, sglv);
    }
  }
  // the endomorphism split ("glv": the caller vouches that the points lie in G1): 2n records, half the windows.  Not for resident
  // vectors (their tables hold n records) nor beyond the partition sort's 2^23 entries per window.
  const bool glv = src.kind != PtSrc::PREPARED && ctx->use_partition_sort && 2 * n <= PART_MAX_N &&
                   (ctx->glv == 2 || (ctx->glv == 1 && n <= (size_t)ctx->glv_max_n));
  if (c == 0) c = pick_plan_c(glv ? 2 * n : n, ctx->auto_plan);
  const int cabs = c < 0 ? -c : c;
  if (cabs < 4 || cabs > 16) { snprintf(ctx->err, sizeof ctx->err, "window width %d out of range [4,16]", c); return CG1_ERR_ARG; }
  const WinPlan plan = make_plan(c, glv);
  ctx->pend_c = c;
  ctx->pend_split = false;
  if (ctx->split && n >= ctx->split_min_n && win_count(plan.nwin, rank, world) >= 2) return msm_begin_split(ctx, src, d_scalars32, n, plan, rank, world);
  return msm_enqueue(ctx, src, d_scalars32, n, plan, rank, world);
}
int msm_end(Ctx* ctx, cg1h::jac& result) {
  const bool was_small = ctx->pend.active && ctx->pend.hout == ctx->h_small_out;
  int rc = msm_finish(ctx, result);                    // (split: the HIGH windows; their Horner runs while the GPU is still on the low half)
  ctx->last_acc_launches = ctx->pend_split ? 2 : (was_small ? 0 : 1);
  if (ctx->pend_split) {
    ctx->pend_split = false;
    Ctx* ch = child_of(ctx);
    const float acc_hi = ctx->phase_ms[4], wait_hi = ctx->host_ms[1], tail_hi = ctx->host_ms[3], enq = ctx->host_ms[0];
    const uint32_t e_hi = ctx->last_entries, c_hi = ctx->last_chunks;
    cg1h::jac lo;
    int rc2 = msm_finish(ch, lo);
    if (rc == CG1_OK) rc = rc2;
    if (rc2 != CG1_OK) snprintf(ctx->err, sizeof ctx->err, "%s", ch->err);
    if (rc == CG1_OK) result = cg1h::jac_add(result, lo);
    // the call's figures: both k_accumulate launches, both chains' entries; host: enqueue of both chains, waits, tails
    for (int i = 0; i < CG1_NPHASE; ++i) ctx->phase_ms[i] += ch->phase_ms[i];
    ctx->phase_ms[4] = acc_hi + ch->phase_ms[4];
    ctx->last_entries = e_hi + ch->last_entries; ctx->last_chunks = c_hi + ch->last_chunks;
    ctx->host_ms[0] = enq + ch->host_ms[0]; ctx->host_ms[1] = wait_hi + ch->host_ms[1]; ctx->host_ms[3] = tail_hi + ch->host_ms[3];
    ctx->host_tail_ms = ch->host_ms[3];                 // what is left on the critical path after the GPU is done
  }
  if (ctx->pend_c) ctx->last_c = ctx->pend_c;          // negative: a balanced plan (cg1_get_timings reports it)
  return rc;
}
int msm_device(Ctx* ctx, const PtSrc& src, const void* d_scalars32, size_t n, int c, int rank, int world, cg1h::jac& result) {
  result = cg1h::jac_identity();
  int rc = msm_begin(ctx, src, d_scalars32, n, c, rank, world);
  if (rc) return rc;
  return msm_end(ctx, result);
}


int pick_window_batched(size_t n_avg, bool glv) {
  const int bits = glv ? 127 : 255;
  if (glv) n_avg *= 2;                                 // entries per MSM: both halves of every term
  int best = 4; double best_cost = 1e300;
  for (int c = 4; c <= 9; ++c) {                       // NB <= 256: a group's counting sort fits one block's LDS
    if (bits % c == 0) continue;                       // top window would hold only the recoding carry: one hot bucket
    int nwin = bits / c + 1;
    double NB = (double)(1u << (c - 1));
    double cost = (double)nwin * ((double)n_avg + 1.4 * (2.0 * NB + 3.0 * NB / 8.0)) + 1.4 * (double)bits;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}
int pick_window_batched(size_t n_avg) { return pick_window_batched(n_avg, false); }

static int ensure_child(Ctx* ctx);      // the second launch chain's context and the two events between the chains (capi_core_msm.h)

// The regime-B launch chain of one (sub-)batch, in two steps so that two of them can be in flight (msm_batched_device below):
// batched_chain_enqueue queues everything up to the D2H of the results and returns; batched_chain_finish waits and reads them back.
struct BatchedChain {
  size_t M = 0; int c = 0; uint32_t nwin = 0; bool host_horner = false;
  std::chrono::steady_clock::time_point h0, h1;
};
static int batched_chain_enqueue(Ctx* ctx, const void* d_points96, const void* d_scalars32, const uint32_t* h_offsets, size_t M, int c, bool glv,
                                 hipEvent_t wait_before_accumulate, hipEvent_t record_after_accumulate, BatchedChain& bc) {
  // regime B with the endomorphism split ("glv" != 0: the caller vouches for G1): 2N records and entries, half the windows -- half the
  // (MSM, window) groups whose buckets k_seg_reduce / k_group_reduce sum at the lane rate, half the doublings of every MSM's Horner
  const size_t N = h_offsets[M], N_real = N;
  HIPCHK(hipSetDevice(ctx->device));
  const WinPlan bplan = make_plan(c, glv);
  const uint32_t nwin = (uint32_t)bplan.nwin, NB = 1u << (c - 1);
  const size_t G = M * nwin, nb_total = G * NB;
  const size_t Nv = glv ? 2 * N : N;                    // entries per digit row = prepared records
  if (nb_total >= (1ull << 31) || Nv * nwin >= (1ull << 31)) { snprintf(ctx->err, sizeof ctx->err, "batch too large"); return CG1_ERR_ARG; }
  const uint32_t m = 8 < NB ? 8 : NB;                 // segment length of k_seg_reduce
  uint32_t log2m = 0; while ((1u << log2m) < m) ++log2m;
  const uint32_t J = NB / m;
  uint32_t L0 = ctx->L0;
  while (L0 < 65536u && ((uint64_t)Nv * (uint64_t)nwin >> 18) > (uint64_t)L0) L0 <<= 1;
  int rc = ensure(ctx, Nv, nb_total, nwin, 1, L0);
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
    HIPCHK(hipMalloc(&ctx->d_bout, (M + 1) * sizeof(PointWords)));         // + one record: the input-validation flag word
    HIPCHK(hipHostMalloc(&ctx->h_bout, (M + 1) * sizeof(PointWords)));
    ctx->cap_bout = M;
  }
  if (Nv * nwin > ctx->cap_digits) {
    if (ctx->d_digits) (void)hipFree(ctx->d_digits);
    HIPCHK(hipMalloc(&ctx->d_digits, Nv * nwin * 2 + 16));
    ctx->cap_digits = Nv * nwin;
  }
  hipStream_t st = ctx->stream;
  const uint32_t N32 = (uint32_t)N_real, gn = (N32 + 255) / 256, Nv32 = (uint32_t)Nv, split = glv ? N32 : 0u;
  auto h0 = std::chrono::steady_clock::now();
  HIPCHK(hipMemcpyAsync(ctx->d_boffs, h_offsets, (M + 1) * 4, hipMemcpyHostToDevice, st));
  if (ctx->profile >= 2) HIPCHK(hipEventRecord(ctx->ev[0], st));
  uint32_t* bad_flag = reinterpret_cast<uint32_t*>(ctx->d_bout + M);
  hipLaunchKernelGGL(k_prepare_points, dim3(gn), dim3(256), 0, st, (const uint32_t*)d_points96, ctx->d_pts, ctx->d_flags, N32, bad_flag);
  if (glv) hipLaunchKernelGGL(k_phi_records, dim3(gn), dim3(256), 0, st, ctx->d_pts, ctx->d_flags, N32);
  if (ctx->profile >= 2) HIPCHK(hipEventRecord(ctx->ev[1], st));
  hipLaunchKernelGGL(k_digits, dim3(gn), dim3(256), 0, st, (const uint32_t*)d_scalars32, ctx->d_flags, ctx->d_digits, N32, bplan, 0, 1, bad_flag);
  hipLaunchKernelGGL(k_group_count, dim3((uint32_t)M, nwin), dim3(256), 0, st, ctx->d_digits, ctx->d_boffs, ctx->d_hist, Nv32, NB, nwin, split);
  if (ctx->profile >= 2) HIPCHK(hipEventRecord(ctx->ev[2], st));
  const uint32_t nblk = (uint32_t)((nb_total + SCAN_ITEMS - 1) / SCAN_ITEMS);
  hipLaunchKernelGGL(k_scan1, dim3(nblk), dim3(256), 0, st, ctx->d_hist, ctx->d_off, ctx->d_choff, ctx->d_blocktot, (uint32_t)nb_total, L0);
  hipLaunchKernelGGL(k_scan2, dim3(1), dim3(256), 0, st, ctx->d_blocktot, nblk, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
  hipLaunchKernelGGL(k_scan3, dim3(nblk), dim3(256), 0, st, ctx->d_blocktot, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
  hipLaunchKernelGGL(k_group_scatter, dim3((uint32_t)M, nwin), dim3(256), 0, st, ctx->d_digits, ctx->d_boffs, ctx->d_off, ctx->d_sorted, Nv32, NB, nwin, split);
  if (ctx->profile >= 2) HIPCHK(hipEventRecord(ctx->ev[3], st));
  HIPCHK(hipMemsetAsync(ctx->d_zblock, 0, zblock_clear_bytes(ctx), st));
  hipLaunchKernelGGL(k_chunk_desc, dim3((uint32_t)std::min<size_t>(CHUNK_DESC_BLOCKS, (nb_total + 255) / 256)), dim3(256), 0, st, ctx->d_off, ctx->d_choff, ctx->d_desc, ctx->d_lenhist, ctx->d_heavy, (uint32_t)ctx->cap_heavy, (uint32_t)nb_total, L0, ctx->d_any_multi);
  const size_t max_chunks = nb_total + (Nv * (size_t)nwin) / L0 + 1;
  const uint32_t gchunks = (uint32_t)((max_chunks + 255) / 256);
  hipLaunchKernelGGL(k_len_scan, dim3(1), dim3(256), 0, st, ctx->d_lenhist, ctx->d_lenhist + LEN_BINS, ctx->d_off + nb_total, ctx->d_choff + nb_total, bad_flag);
  hipLaunchKernelGGL(k_order, dim3((gchunks + ORDER_PER - 1) / ORDER_PER), dim3(256), 0, st, ctx->d_desc, ctx->d_choff + nb_total, ctx->d_lenhist + LEN_BINS, ctx->d_order);
  if (wait_before_accumulate) HIPCHK(hipStreamWaitEvent(st, wait_before_accumulate, 0));
  if (ctx->profile >= 1) HIPCHK(hipEventRecord(ctx->ev[4], st));
  hipLaunchKernelGGL(k_accumulate, dim3(gchunks), dim3(256), 0, st, ctx->d_desc, ctx->d_choff + nb_total, ctx->d_order, ctx->d_sorted, ctx->d_pts, ctx->d_sums);
  if (ctx->profile >= 1) HIPCHK(hipEventRecord(ctx->ev[5], st));
  if (record_after_accumulate) HIPCHK(hipEventRecord(record_after_accumulate, st));
  const uint32_t nseg_total = (uint32_t)(nb_total / m);
  hipLaunchKernelGGL(k_heavy_combine, dim3(512), dim3(256), 0, st, ctx->d_heavy, (uint32_t)ctx->cap_heavy, ctx->d_choff, ctx->d_sums, ctx->d_combined);
  hipLaunchKernelGGL(k_seg_reduce, dim3((nseg_total + 255) / 256), dim3(256), 0, st, ctx->d_choff, ctx->d_sums, ctx->d_combined, ctx->d_segrun, ctx->d_segtot, nseg_total, m);
  hipLaunchKernelGGL(k_group_reduce, dim3((uint32_t)((G + 255) / 256)), dim3(256), 0, st, ctx->d_segrun, ctx->d_segtot, ctx->d_gsum, (uint32_t)G, J, log2m);
  if (ctx->profile >= 2) HIPCHK(hipEventRecord(ctx->ev[6], st));
  // The Horner over an MSM's window sums is 255 DEPENDENT doublings: ~1.0 ms for a DPP quad, ~65 us for a host core.  A handful of
  // MSMs (the prover's halving rounds: 4-6 per call) therefore finish on the host, on the context's four threads; hundreds of them
  // (a batch of accumulator MSMs) keep the device's one-quad-per-MSM kernel, which does them all in the same millisecond.
  const bool host_horner = M <= (size_t)ctx->batched_host_horner_max;
  if (host_horner) {
    if (G > ctx->cap_gout) {
      if (ctx->d_gout) (void)hipFree(ctx->d_gout);
      if (ctx->h_gout) (void)hipHostFree(ctx->h_gout);
      ctx->d_gout = nullptr; ctx->h_gout = nullptr; ctx->cap_gout = 0;
      HIPCHK(hipMalloc(&ctx->d_gout, G * sizeof(PointWords)));
      HIPCHK(hipHostMalloc(&ctx->h_gout, G * sizeof(PointWords)));
      ctx->cap_gout = G;
    }
    hipLaunchKernelGGL(k_export_sums, dim3((uint32_t)((G + 63) / 64)), dim3(64), 0, st, ctx->d_gsum, ctx->d_gout, (uint32_t)G);
    HIPCHK(hipMemcpyAsync(ctx->h_gout, ctx->d_gout, G * sizeof(PointWords), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(ctx->h_bout + M, ctx->d_bout + M, sizeof(PointWords), hipMemcpyDeviceToHost, st));      // the status words
  } else {
    // up to ~2 000 MSMs one WAVE each, one limb per lane (fp_row.h: a lone wave's doubling in ~1.5 us instead of a quad's ~5); beyond,
    // a wave per MSM would be eight and more to a SIMD and the quads' throughput wins (profiles/r05_rowlane_ab.txt)
    if (ctx->horner_row && M <= 2048) hipLaunchKernelGGL(k_msm_horner_row, dim3((uint32_t)M), dim3(64), 0, st, ctx->d_gsum, ctx->d_bout, (uint32_t)M, nwin, (uint32_t)c);
    else if (ctx->quad) hipLaunchKernelGGL(k_msm_horner_quad, dim3((uint32_t)((M * 4 + 63) / 64)), dim3(64), 0, st, ctx->d_gsum, ctx->d_bout, (uint32_t)M, nwin, (uint32_t)c);
    else hipLaunchKernelGGL(k_msm_horner, dim3((uint32_t)((M + 63) / 64)), dim3(64), 0, st, ctx->d_gsum, ctx->d_bout, (uint32_t)M, nwin, (uint32_t)c);
    HIPCHK(hipMemcpyAsync(ctx->h_bout, ctx->d_bout, (M + 1) * sizeof(PointWords), hipMemcpyDeviceToHost, st));
  }
  if (ctx->profile >= 2) HIPCHK(hipEventRecord(ctx->ev[7], st));
  bc.M = M; bc.c = c; bc.nwin = nwin; bc.host_horner = host_horner; bc.h0 = h0; bc.h1 = std::chrono::steady_clock::now();
  return CG1_OK;
}

static int batched_chain_finish(Ctx* ctx, const BatchedChain& bc, cg1h::jac* results) {
  const size_t M = bc.M; const int c = bc.c; const uint32_t nwin = bc.nwin; const bool host_horner = bc.host_horner;
  const auto h0 = bc.h0, h1 = bc.h1;
  HIPCHK(hipSetDevice(ctx->device));
  { int wrc = wait_stream(ctx); if (wrc) return wrc; }
  HIPCHK(hipGetLastError());
  {
    const uint32_t* st_words = reinterpret_cast<const uint32_t*>(ctx->h_bout + M);
    ctx->last_entries = st_words[1]; ctx->last_chunks = st_words[2];
    if (st_words[0]) {
      snprintf(ctx->err, sizeof ctx->err, "a scalar is >= 2^255: scalar32 must be a canonical Fr element (< r)");
      return CG1_ERR_ENCODING;
    }
  }
  auto h2 = std::chrono::steady_clock::now();
  { int erc = read_phase_events(ctx, ctx->profile); if (erc) return erc; }
  ctx->last_c = c;
  if (host_horner) {
    auto one = [&](size_t j) {
      cg1h::jac a = cg1h::jac_identity();
      for (int w = (int)nwin - 1; w >= 0; --w) {
        for (int k = 0; k < c; ++k) a = cg1h::jac_dbl(a);
        a = cg1h::jac_add(a, jac_from_words(ctx->h_gout[j * nwin + (size_t)w]));
      }
      results[j] = a;
    };
    const size_t nth = std::min<size_t>(4, M);
    for (size_t t = 1; t < nth; ++t) ctx->helper[t - 1].run([&, t]() { for (size_t j = t; j < M; j += nth) one(j); });
    for (size_t j = 0; j < M; j += nth) one(j);
    for (size_t t = 1; t < nth; ++t) ctx->helper[t - 1].wait();
  } else {
    for (size_t j = 0; j < M; ++j) results[j] = jac_from_words(ctx->h_bout[j]);
  }
  auto h3 = std::chrono::steady_clock::now();
  ctx->host_ms[0] = std::chrono::duration<float, std::milli>(h1 - h0).count();
  ctx->host_ms[1] = std::chrono::duration<float, std::milli>(h2 - h1).count();
  ctx->host_ms[2] = 0;
  ctx->host_ms[3] = ctx->host_tail_ms = std::chrono::duration<float, std::milli>(h3 - h2).count();
  return CG1_OK;
}

// M independent MSMs over one concatenated (points, scalars) input resident on the device.
// async_small (may be NULL): when the call fits ONE k_msm_small launch it is only ENQUEUED and *async_small set; the caller does other
// work and collects the results with msm_batched_small_end.  Calls that take the regime-B chain complete before returning.
static int msm_batched_small_end(Ctx* ctx, size_t M, std::vector<cg1h::jac>& results) {
  if (M == 1) { cg1h::jac r; int rc = msm_finish(ctx, r); ctx->last_acc_launches = 0; if (rc == CG1_OK) results[0] = r; return rc; }
  return msm_small_batched_finish(ctx, (uint32_t)M, results);
}
// d_offsets_ready (may be NULL): the same M + 1 offsets at an address the DEVICE can read (mapped host memory); the single-launch path
// then reads them there instead of waiting for a copy.
int msm_batched_device(Ctx* ctx, const void* d_points96, const void* d_scalars32, const uint32_t* h_offsets, size_t M,
                       int c, std::vector<cg1h::jac>& results, bool* async_small = nullptr, const uint32_t* d_offsets_ready = nullptr) {
  if (async_small) *async_small = false;
  results.assign(M, cg1h::jac_identity());
  if (M == 0) return CG1_OK;
  const size_t N = h_offsets[M];
  for (size_t j = 0; j < M; ++j) if (h_offsets[j] > h_offsets[j + 1]) { snprintf(ctx->err, sizeof ctx->err, "offsets not monotone"); return CG1_ERR_ARG; }
  if (h_offsets[0] != 0) { snprintf(ctx->err, sizeof ctx->err, "offsets[0] must be 0"); return CG1_ERR_ARG; }
  if (N == 0) return CG1_OK;
  if (N >= (1ull << 31) || M > 65535) { snprintf(ctx->err, sizeof ctx->err, "batch too large"); return CG1_ERR_ARG; }
  {
    // A handful of small MSMs (the 4 - 6 of a prover's halving round, prover_kernels.py): ONE k_msm_small launch carries them all
    // (grid.z = MSM) and their Horners run side by side on the host -- the regime-B launch chain costs ~0.5 ms whatever it sums.
    size_t max_n = 0;
    for (size_t j = 0; j < M; ++j) max_n = std::max<size_t>(max_n, h_offsets[j + 1] - h_offsets[j]);
    const bool sglv = ctx->glv && 2 * max_n <= SM_MAX_N;
    const size_t max_nn = sglv ? 2 * max_n : max_n;
    const int cs = c > 0 ? c : pick_small_c(max_nn);
    const size_t groups = (size_t)M * (size_t)((sglv ? 127 : 255) / cs + 1) * ((max_nn + SM_SLICE - 1) / SM_SLICE);
    if (ctx->small_msm && M <= SM_MAX_MSMS && max_nn <= SM_MAX_N && groups <= SM_MAX_GROUPS && cs >= 4 && cs <= 9 && cs != 5) {
      HIPCHK(hipSetDevice(ctx->device));
      const uint32_t* d_offs = d_offsets_ready;
      if (!d_offs) {
        if ((M + 1) > ctx->cap_boffs) {
          if (ctx->d_boffs) (void)hipFree(ctx->d_boffs);
          ctx->d_boffs = nullptr; ctx->cap_boffs = 0;
          HIPCHK(hipMalloc(&ctx->d_boffs, (M + 1) * 4));
          ctx->cap_boffs = M + 1;
        }
        HIPCHK(hipMemcpyAsync(ctx->d_boffs, h_offsets, (M + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        d_offs = ctx->d_boffs;
      }
      int rc = msm_enqueue_small(ctx, PtSrc(d_points96), d_scalars32, N, cs, (uint32_t)M, d_offs, max_n, sglv);
      if (rc) return rc;
      if (async_small) { *async_small = true; return CG1_OK; }
      return msm_batched_small_end(ctx, M, results);
    }
  }
  // regime B with the endomorphism split ("glv" != 0: the caller vouches for G1): 2N records and entries, half the windows -- half the
  // (MSM, window) groups whose buckets k_seg_reduce / k_group_reduce sum at the lane rate, half the doublings of every MSM's Horner
  const bool glv = ctx->glv != 0 && 2 * N < (1ull << 31);
  if (c <= 0) c = pick_window_batched((N + M - 1) / M, glv);
  if (c < 4 || c > 9) { snprintf(ctx->err, sizeof ctx->err, "batched window width %d out of range [4,9]", c); return CG1_ERR_ARG; }
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->batched_split || M < (size_t)ctx->batched_split_min_m || h_offsets[M / 2] == 0 || h_offsets[M / 2] == N) {      // (an empty half: one chain)
    BatchedChain bc;
    int rc = batched_chain_enqueue(ctx, d_points96, d_scalars32, h_offsets, M, c, glv, nullptr, nullptr, bc);
    if (rc) return rc;
    return batched_chain_finish(ctx, bc, results.data());
  }
  // Two chains, half the MSMs each, on two streams: the second half's k_accumulate starts when the first half's has finished, so the
  // first half's latency-bound tail (group sums, one wave per MSM for 255 doublings, D2H) runs UNDER the second half's additions
  // instead of after them.  ("batched_split", default off: measured slower, profiles/r05_regime_b_split.txt.)
  { int crc = ensure_child(ctx); if (crc) return crc; }
  Ctx* ch = child_of(ctx);
  ch->profile = ctx->profile; ch->L0 = ctx->L0; ch->quad = ctx->quad; ch->horner_row = ctx->horner_row; ch->blocking_sync = ctx->blocking_sync;
  ch->batched_host_horner_max = ctx->batched_host_horner_max; ch->use_partition_sort = ctx->use_partition_sort;
  const size_t Mlo = M / 2, Mhi = M - Mlo;
  std::vector<uint32_t> offs_hi(Mhi + 1);
  for (size_t j = 0; j <= Mhi; ++j) offs_hi[j] = h_offsets[Mlo + j] - h_offsets[Mlo];
  BatchedChain lo, hi;
  int rc = batched_chain_enqueue(ctx, d_points96, d_scalars32, h_offsets, Mlo, c, glv, nullptr, ctx->ev_acc, lo);
  if (rc) return rc;
  rc = batched_chain_enqueue(ch, static_cast<const uint8_t*>(d_points96) + 96ull * h_offsets[Mlo], static_cast<const uint8_t*>(d_scalars32) + 32ull * h_offsets[Mlo],
                             offs_hi.data(), Mhi, c, glv, ctx->ev_acc, nullptr, hi);
  if (rc) { snprintf(ctx->err, sizeof ctx->err, "%s", ch->err); (void)batched_chain_finish(ctx, lo, results.data()); return rc; }
  rc = batched_chain_finish(ctx, lo, results.data());
  const float acc_lo = ctx->phase_ms[4], enq_lo = ctx->host_ms[0], wait_lo = ctx->host_ms[1];
  const uint32_t e_lo = ctx->last_entries, c_lo = ctx->last_chunks;
  int rc2 = batched_chain_finish(ch, hi, results.data() + Mlo);
  if (rc2 != CG1_OK) snprintf(ctx->err, sizeof ctx->err, "%s", ch->err);
  if (rc == CG1_OK) rc = rc2;
  ctx->phase_ms[4] = acc_lo + ch->phase_ms[4];
  ctx->last_entries = e_lo + ch->last_entries; ctx->last_chunks = c_lo + ch->last_chunks;
  ctx->host_ms[0] = enq_lo + ch->host_ms[0]; ctx->host_ms[1] = wait_lo + ch->host_ms[1];
  return rc;
}

}  // namespace cg1
