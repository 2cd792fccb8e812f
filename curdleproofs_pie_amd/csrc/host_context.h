// Host side of the MSM engine, part 1: the per-device context (streams, helper threads, scratch buffers and their growth).
// Part of the single translation unit csrc/msm_gpu.hip (included there, in this order; not a stand-alone header).
#pragma once

namespace cg1 {

// ------------------------------------------------------------------ host-side context
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  snprintf(ctx->err, sizeof ctx->err, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); return CG1_ERR_HIP; } } while (0)

// One persistent helper thread per context for the second half of the host Horner tail (a std::async per call paid a
// thread creation, ~40 us, on a ~150 us tail).
struct Helper {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has = false, done = true, quit = false, armed = false;
  std::atomic<bool> posted{false};       // mirrors `has` for a helper that is spinning (arm())
  std::atomic<bool> stand_down{false};   // disarm(): the spinning helper gives up at once
  void start_locked() { if (!th.joinable()) th = std::thread([this]() { loop(); }); }
  void run(std::function<void()> f) {
    std::unique_lock<std::mutex> lk(mu);
    start_locked();
    job = std::move(f); has = true; done = false;
    posted.store(true, std::memory_order_release);
    cv.notify_all();
  }
  // A job is about to come (the caller starts polling for a GPU result a fraction of a millisecond away): wake the thread now and
  // let it SPIN for the job (at most ~2 ms) instead of paying the futex wake-up -- 20-40 us -- inside a 100 us host tail.
  void arm() {
    std::unique_lock<std::mutex> lk(mu);
    start_locked();
    if (!has && done) { stand_down.store(false, std::memory_order_relaxed); armed = true; cv.notify_all(); }
  }
  // no job will come after all (the call failed, or its tail needs fewer threads than were armed): stop spinning, go back to sleep
  void disarm() { stand_down.store(true, std::memory_order_release); }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this]() { return done; });
  }
  void loop() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [this]() { return has || quit || armed; });
      if (quit) return;
      if (!has) {                                              // armed: spin for the job outside the lock
        armed = false;
        lk.unlock();
        const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(2);
        for (uint32_t k = 0; !posted.load(std::memory_order_acquire); ++k) {
          if (stand_down.load(std::memory_order_acquire)) break;
          if ((k & 0xffu) == 0xffu && std::chrono::steady_clock::now() > until) break;
          __builtin_ia32_pause();
        }
        lk.lock();
        if (!has) continue;
      }
      armed = false;
      std::function<void()> f = std::move(job);
      has = false;
      posted.store(false, std::memory_order_relaxed);
      lk.unlock();
      f();
      lk.lock();
      done = true;
      cv.notify_all();
    }
  }
  ~Helper() {
    { std::unique_lock<std::mutex> lk(mu); quit = true; cv.notify_all(); }
    if (th.joinable()) th.join();
  }
};

struct Ctx {
  int device = 0;
  Helper helper[3];                     // the host Horner tail runs on up to four threads (this one + three helpers)
  hipStream_t stream = nullptr;
  std::atomic<hipStream_t> copy_stream{nullptr};    // H2D staging copies that overlap kernels on `stream` (cg1_h2d_async / cg1_copy_fence); created at the first
  std::once_flag copy_once;              // such copy: a context that never stages (the verifier's front-end lanes) holds ONE stream -- a process
                                        // has 24 hardware queues, and streams that share one run one after the other
  hipEvent_t copy_ev = nullptr;
  std::vector<uint32_t> cu_mask;        // non-empty: the compute and side streams are confined to these CUs
  hipStream_t side_stream = nullptr;    // small latency-bound kernels that run BESIDE the compute stream (cg1_subgroup_flags_enqueue)
  hipEvent_t side_ev = nullptr;
  hipEvent_t sync_ev = nullptr;         // blocking-sync event: waits sleep on an interrupt instead of spinning a core
  int blocking_sync = 0;
  char err[256] = {0};
  // capacity
  size_t cap_n = 0, cap_nb = 0, cap_chunks = 0, cap_entries = 0, cap_out = 0;
  PreparedPoint* d_pts = nullptr;
  uint8_t* d_flags = nullptr;
  uint32_t *d_hist = nullptr, *d_off = nullptr, *d_choff = nullptr, *d_sorted = nullptr;
  uint2 *d_blocktot = nullptr, *d_desc = nullptr;
  uint32_t *d_order = nullptr, *d_lenhist = nullptr;      // [2*LEN_BINS]: histogram, cursor  (points into d_zblock)
  // one block cleared by ONE memset per call: chunk-length histogram | any_multi flag | combined[] bytes | heavy count (+ ids)
  uint32_t* d_zblock = nullptr; size_t cap_zblock = 0;
  uint32_t* d_any_multi = nullptr;
  PointSum* d_partial = nullptr; size_t cap_partial = 0;
  uint16_t* d_digits = nullptr; uint32_t* d_part = nullptr; uint32_t* d_blockcnt = nullptr; uint32_t* d_ublocktot = nullptr;
  size_t cap_digits = 0, cap_part = 0, cap_blockcnt = 0;
  uint32_t *d_slice_base = nullptr, *d_slicehist = nullptr, *d_subbase = nullptr; uint8_t* d_bigflag = nullptr;
  size_t cap_bigflag = 0, cap_slices = 0;
  int big_bins = 1;                     // giant bins of skewed scalars sorted by many blocks (A/B switch)
  int use_partition_sort = 1;
  int stage_sort = 1;                   // LDS-staged, line-coalesced writes in k_part_scatter / k_bin_sort (A/B switch)
  int host_split = 1;                   // host Horner tail on two threads (A/B switch)
  int rowcol_quad = 1;                  // k_rowcol_quad for small bucket counts (A/B switch)
  int rowcol_lgq = 0;                   // "rowcol_lgq": log2 of the quads per row / column of k_rowcol_quad (2, 3, 4; 0 = by cost)
  int rowcol_quad_max = 1 << 18;        // ... up to this many buckets ("rowcol_quad_max")
  int tree_shift = 2;                   // "tree_shift": k_small_tree_quad's block = 4 lanes per element >> this (0 .. 4; -1 = tree_half's 0 / 1).  Measured
                                        // (profiles/r04_tree_ab.txt, tree + export at 2^16 / 2^18 / 2^20): 0: 111 / 112 / 119 us, 1: 90 / 112 / 121, 2: 79 / 100 / 110, 4: 78 / 161 / 177
  int tree_half = 1;                    // k_small_tree_quad: 2 lanes per element (a quad takes two elements) instead of 4 (A/B switch)
  int merlin_sync = 1;                  // k_merlin_batch_sync (lanes permute together) instead of k_merlin_batch (A/B switch)
  uint32_t merlin_clk[2] = {0, 0};
  int fe_timed = 0;                     // "fe_timed": the block-program kernel reads the shader clock around the parts of a pass (cg1_shuffle_fe_last_split)
  int fe_rows = 1;                      // "fe_rows": 1 = the front-end's block-program kernel (k_shuffle_front_end_rows), 0 = the byte machine
  int fe_prio = 0;                      // "fe_prio": wave priority of k_shuffle_front_end (s_setprio 0 .. 3)
  int decompress_waves = 3;             // "decompress_waves": waves per SIMD k_batch_decompress<false> is compiled for (2: table in registers, 3: half of it in scratch)
  void* d_opening = nullptr; size_t cap_opening = 0;              // cg1_opening_prepare_device's scratch (940 B per proof)
  int merlin_last_kernel = 0;           // which kernel served the last cg1_merlin_batch_device call: 2 block program, 1 byte machine, 0 one lane at a time
  int merlin_rows = 1;                  // "merlin_rows": 1 = cg1_merlin_batch_device hashes whole rate blocks (k_merlin_batch_rows) when the program fits, 0 = byte machine
  void* d_merlin_rows = nullptr; size_t merlin_rows_cap = 0;      // the rows of the last such call (kept: 134 KB per shuffle-shaped transcript)
  int merlin_lanes = 64;                // transcripts per wave of k_merlin_batch_sync ("merlin_lanes": 1 .. 64)
  uint32_t merlin_passes = 0;           // of the last cg1_merlin_batch_device call: Keccak passes of the slowest wave
  int sort_sub_bits = 0;                // partition sort: sub-bucket bits a k_bin_sort workgroup sorts by ("sort_sub_bits": 4 .. 8; 0 = 7 up to 2^16 terms, else 8)
  int batch_mul_host_max = -1;          // cg1_batch_mul_add (host pointers): outputs up to which the host's pool does the work ("batch_mul_host_max"; -1 = 16 per pool thread, 0 = never)
  int last_batch_mul_on_host = 0;
  int batch_mul_quad_max = 8192;        // k_batch_mul_quad up to this many outputs ("batch_mul_quad_max"; 0 = always one lane per output)
  int scan_one = 1;                     // the sort's two scans as one single-block launch each when they are small (A/B switch)
  int fold_pass = 1;                    // k_bucket_fold in front of k_rowcol / k_seg_reduce; 0 leaves multi-chunk buckets to their bucket_sum loops
                                        // (measured WORSE: 372 instead of 235 us at 2^16 -- divergent trip counts inside the row / column lanes)
  int auto_plan = 1;                    // window_c = 0 picks balanced window plans for mid-size inputs (A/B switch)
  struct Pending {                      // what msm_finish needs from msm_enqueue
    bool active = false;
    int c = 0, rank = 0, world = 1, nlw = 0, nbits = 0;
    WinPlan plan;
    uint32_t m = 1, lb2 = 0, hb2 = 0, nitems = 0;
    bool use2d = true;
    int profile = 0;                    // the level the events of THIS call were recorded under (may change before msm_finish)
    bool zero_copy = false; uint32_t seq = 0;
    bool arm_helpers = false;           // the host tail is a large share of this call: its helper threads spin for their part while the GPU result is polled
    size_t nout_words = 0;
    const PointWords* hout = nullptr;   // where the exported items land (ctx->h_out, or h_small_out for k_msm_small)
    std::chrono::steady_clock::time_point h0, h1;
  } pend;
  // "split" (A/B switch, OFF): one large call as TWO launch chains on two streams -- the high half of the windows on this context, the
  // low half on `child` (own scratch buffers, stream and export flag; shared prepared points) -- meant to run the low half's sort under
  // the high half's k_accumulate and the high half's reduction tail under the low half's.  MEASURED A LOSS (profiles/r04_split_ab.txt:
  // 2^20 3.25 ms against 2.95, 2^18 1.54 against 1.22; only 2^16 gains 3 %): the resident blocks of k_accumulate hold every SIMD's
  // registers for their whole ~1 ms life, so the other stream's kernels are dispatched only when it drains -- the two chains run one
  // after the other, and each pays its own launch chain and the shorter chunks of half the entries.
  int split = 0;
  size_t split_min_n = (size_t)1 << 17;
  cg1_ctx* child = nullptr;
  hipEvent_t ev_prep = nullptr, ev_acc = nullptr;
  bool pend_split = false;
  int last_acc_launches = 0;            // k_accumulate launches of the last MSM call: 2 (split), 1, or 0 (k_msm_small)
  int small_row_tail = 1;               // "small_row_tail": k_msm_small's items / combine / export one limb per lane, one wave per item (A/B switch)
  int small_msm = 1;                    // "small_msm": MSMs of <= SM_MAX_N = 2048 terms as ONE launch (k_msm_small); 0 = the regime-A chain (A/B switch)
  PointSum* d_small_partial = nullptr; size_t cap_small_partial = 0;
  uint32_t* d_small_ctr = nullptr;
  PreparedPoint* d_small_pts = nullptr; uint8_t* d_small_flags = nullptr; size_t cap_small_pts = 0;      // k_prepare_blobs<true> output for un-normalised blob input
  PointWords* h_small_out = nullptr; PointWords* h_small_out_dev = nullptr;     // pinned + mapped: 64 x 9 window items + the status record
  int quad = 1;                         // quad-lane EC ops in the latency-bound kernels (A/B switch)
  int reduce_2d = 1;                    // 1: k_rowcol + k_small_tree; 0: k_seg_reduce + k_bit_tree (A/B switch)
  uint32_t* d_heavy = nullptr; size_t cap_heavy = 0;         // [0] count, then heavy bucket ids
  uint8_t* d_combined = nullptr; size_t cap_combined = 0;
  uint32_t* d_boffs = nullptr; size_t cap_boffs = 0;          // regime B: MSM offsets, group sums, per-MSM results
  PointSum* d_gsum = nullptr; size_t cap_gsum = 0;
  PointWords* d_bout = nullptr; PointWords* h_bout = nullptr; size_t cap_bout = 0;
  PointWords* d_gout = nullptr; PointWords* h_gout = nullptr; size_t cap_gout = 0;       // regime B, few MSMs: window sums exported for the host Horner
  int glv = 0;                          // "glv": the caller vouches that every point of its MSM calls lies in the prime-order subgroup, and the engine may run
                                        // the endomorphism split (csrc/glv.h): 1 = where it pays (the single-launch kernel up to 1 024 terms, regime A up to
                                        // glv_max_n terms), 2 = wherever it can (A/B runs).  WRONG results outside G1: default 0.
  int glv_max_n = 1 << 14;              // "glv_max_n": largest regime-A call glv = 1 splits (profiles/r05_glv_ab.txt: slower from 2^15 terms up)
  int lincomb_zero_copy = 1;            // "lincomb_zero_copy": small GPU shares of cg1_lincomb_batch are read from mapped host memory (no staged copy); A/B switch
  int fold_quad = 1;                    // "fold_quad": small bucket counts: k_bucket_fold_quad (1) or the one-lane-per-bucket k_bucket_fold (0); A/B switch
  int rowcol_row = 1;                   // "rowcol_row": with tree_row, small bucket counts: k_rowcol_quad_row (one wave per row / column, the cross-quad levels on rows)
  int tree_row = 1;                     // "tree_row": regime A's 1 + hb + lb items per window by blocks of waves with one limb per lane (k_small_tree_row); 0: k_small_tree_quad
  int horner_row = 1;                   // "horner_row": regime B's device Horner with one wave per MSM, one limb per lane (A/B switch; 0: one quad per MSM)
  int batch_mul_row = 1;                // "batch_mul_row": deferred map / fold batches of 96 .. 4096 results on k_batch_mul_row (A/B switch; 0: pool / k_batch_mul)
  int batched_split = 0;                // "batched_split": regime B as two launch chains of half the MSMs each, the second half's k_accumulate behind the first's
                                        // (so that the first half's tail runs under it).  Measured SLOWER (1 024 x 627: 5.2-5.3 -> 5.6-5.8 ms, with the
                                        // endomorphism split 4.72 -> 4.76-4.85, profiles/r05_regime_b_split.txt): the tails are not idle time.  A/B switch, off.
  int batched_split_min_m = 256;        // "batched_split_min_m": ... from this many MSMs on
  int batched_host_horner_max = 24;     // regime B calls with at most this many MSMs run their Horner on the host ("batched_host_horner_max")
  PointSum *d_sums = nullptr, *d_segrun = nullptr, *d_segtot = nullptr;
  PointWords* d_out = nullptr;
  PointWords* h_out = nullptr;          // pinned, and mapped into the device: k_export_host writes the window sums straight into it
  PointWords* h_out_dev = nullptr;      // the device's address of h_out
  uint32_t* h_flag = nullptr;           // pinned + mapped: k_export_host stores the call's sequence number here when h_out is complete
  uint32_t* h_flag_dev = nullptr;
  uint32_t seq = 0;
  int zero_copy = 1;                    // 1: export kernel + flag polling instead of a D2H copy + stream wait (A/B switch)
  int arm_helpers = 1;                  // "arm_helpers": the Horner's helper threads spin for their part while a small / mid-size call's result is polled (A/B switch)
  int horner_threads = 4;               // host threads of the Horner tail: 1, 2 or 4 (A/B switch; host_split = 0 forces 1)
  // staging for host-pointer entry points
  void* d_stage_pts = nullptr; void* d_stage_sc = nullptr; size_t cap_stage_pts = 0, cap_stage_sc = 0;      // bytes
  uint8_t* h_lin = nullptr; uint8_t* h_lin_dev = nullptr; size_t cap_h_lin = 0;   // page-locked gather buffer of cg1_lincomb_batch (terms' points | scalars)
  // timing
  hipEvent_t ev[CG1_NPHASE + 1];
  float phase_ms[CG1_NPHASE] = {0};
  float host_tail_ms = 0;
  float host_ms[4] = {0, 0, 0, 0};      // enqueue, wait-for-GPU, event readout, Horner tail
  int profile = 1;                      // 0: no hipEvents; 1: around k_accumulate only; 2: around every phase (read_phase_events)
  uint32_t last_chunks = 0, last_entries = 0;   // of the last MSM call: non-zero digits sorted into buckets; chunks k_accumulate ran
  hipEvent_t tm_ev[2] = {nullptr, nullptr};     // cg1_timer_begin / cg1_timer_end
  int last_c = 0, pend_c = 0;
  int chunk_rule = 1;                   // "chunk_rule": whole-bucket chunks at 2^17 .. 2^19 terms (A/B switch)
  uint32_t L0 = 8;                      // MINIMUM chunk length; the per-call length grows with the entry count
  uint32_t seg_m = 4;
};

static void free_bufs(Ctx* c) {
  auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
  F(c->d_pts); F(c->d_flags); F(c->d_hist); F(c->d_off); F(c->d_choff); F(c->d_sorted); F(c->d_blocktot); F(c->d_desc);
  F(c->d_sums); F(c->d_segrun); F(c->d_segtot); F(c->d_out); F(c->d_order); F(c->d_zblock); F(c->d_partial);
  c->d_lenhist = c->d_heavy = c->d_any_multi = nullptr; c->d_combined = nullptr; c->cap_zblock = 0; c->cap_heavy = 0; c->cap_combined = 0;
  c->cap_partial = 0;
  F(c->d_digits); F(c->d_part); F(c->d_blockcnt); F(c->d_ublocktot); F(c->d_boffs); F(c->d_gsum); F(c->d_bout);
  F(c->d_slice_base); F(c->d_slicehist); F(c->d_subbase); F(c->d_bigflag); c->cap_bigflag = 0; c->cap_slices = 0;
  if (c->h_bout) { (void)hipHostFree(c->h_bout); c->h_bout = nullptr; }
  if (c->h_lin) { (void)hipHostFree(c->h_lin); c->h_lin = nullptr; c->h_lin_dev = nullptr; c->cap_h_lin = 0; }
  if (c->d_gout) { (void)hipFree(c->d_gout); c->d_gout = nullptr; }
  if (c->h_gout) { (void)hipHostFree(c->h_gout); c->h_gout = nullptr; }
  c->cap_gout = 0;
  c->cap_boffs = c->cap_gsum = c->cap_bout = 0;
  c->cap_digits = c->cap_part = c->cap_blockcnt = 0;
  if (c->h_out) { (void)hipHostFree(c->h_out); c->h_out = nullptr; }
  c->cap_n = c->cap_nb = c->cap_chunks = c->cap_entries = c->cap_out = 0;
}

static int ensure(Ctx* ctx, size_t n, size_t nb_total, size_t nlw, size_t nitems, uint32_t L, bool need_points = true) {
  size_t entries = n * nlw;
  size_t chunks = nb_total + entries / L + 1;
  if (need_points && n > ctx->cap_n) {
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
  {
    // [lenhist 2*LEN_BINS words][any_multi][combined: nb_total bytes][heavy count][heavy ids]: the call clears everything up to
    // and including the heavy count with one memset
    const size_t hcap = entries / ((size_t)L * (HEAVY_MIN_CHUNKS - 1)) + 2;     // a heavy bucket holds > (MIN-1)*L entries
    const size_t z0 = 2 * LEN_BINS + 1, h0 = z0 + (nb_total + 3) / 4, words = h0 + 1 + hcap + 64;     // (+64: room for the 256-byte round-up of the per-call memset)
    if (words > ctx->cap_zblock) {
      if (ctx->d_zblock) (void)hipFree(ctx->d_zblock);
      ctx->d_zblock = nullptr; ctx->cap_zblock = 0;
      HIPCHK(hipMalloc(&ctx->d_zblock, words * 4));
      ctx->cap_zblock = words;
    }
    ctx->d_lenhist = ctx->d_zblock;
    ctx->d_any_multi = ctx->d_zblock + 2 * LEN_BINS;
    ctx->d_combined = reinterpret_cast<uint8_t*>(ctx->d_zblock + z0);
    ctx->d_heavy = ctx->d_zblock + h0;
    ctx->cap_heavy = hcap;
    ctx->cap_combined = nb_total;
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
    if (nlw * 128 > ctx->cap_bigflag) {
      auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
      F(ctx->d_bigflag); F(ctx->d_slice_base); F(ctx->d_subbase);
      HIPCHK(hipMalloc(&ctx->d_bigflag, nlw * 128 + 16));
      HIPCHK(hipMalloc(&ctx->d_slice_base, (nlw * 128 + 1) * 4));
      HIPCHK(hipMalloc(&ctx->d_subbase, nlw * 128 * 256 * 4));
      ctx->cap_bigflag = nlw * 128;
    }
    {
      const size_t max_slices = entries / SLICE + nlw * 128 + 1;       // sum over bins of ceil(size / SLICE)
      if (max_slices > ctx->cap_slices) {
        if (ctx->d_slicehist) (void)hipFree(ctx->d_slicehist);
        HIPCHK(hipMalloc(&ctx->d_slicehist, max_slices * 256 * 4));
        ctx->cap_slices = max_slices;
      }
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
    HIPCHK(hipMalloc(&ctx->d_out, (nout + 1) * sizeof(PointWords)));      // + one record: the input-validation flag word
    // mapped + coherent, said explicitly: the export kernel writes it and the host polls the flag word without any runtime call in
    // between (with HIP_HOST_COHERENT=0 the default allocation is non-coherent and the poll would only end through its stream query)
    HIPCHK(hipHostMalloc(&ctx->h_out, (nout + 1) * sizeof(PointWords), hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(hipHostGetDevicePointer((void**)&ctx->h_out_dev, ctx->h_out, 0));
    ctx->cap_out = nout;
  }
  return CG1_OK;
}

// bytes of d_zblock the per-call memset clears: everything up to and including the heavy-bucket count, rounded up to 256 B
// (one fill kernel instead of an aligned body + a tail; the heavy ids it may touch are written later by k_chunk_desc)
static size_t zblock_clear_bytes(const Ctx* ctx) {
  size_t bytes = (size_t)((ctx->d_heavy + 1) - ctx->d_zblock) * 4;
  bytes = (bytes + 255) & ~(size_t)255;
  const size_t cap = ctx->cap_zblock * 4;
  return bytes < cap ? bytes : cap;
}

// profile 2: every phase is bracketed by hipEvents; 1 (default): only k_accumulate (the roofline kernel) -- each event record
// is a marker packet that costs the stream ~5.5 us, 8 of them were 4 % of a 2^16-term MSM; 0: none.
static int read_phase_events(Ctx* ctx, int profile) {
  for (int i = 0; i < CG1_NPHASE; ++i) ctx->phase_ms[i] = 0.f;
  if (profile >= 2) {
    for (int i = 0; i < CG1_NPHASE; ++i) HIPCHK(hipEventElapsedTime(&ctx->phase_ms[i], ctx->ev[i], ctx->ev[i + 1]));
  } else if (profile == 1) {
    HIPCHK(hipEventElapsedTime(&ctx->phase_ms[4], ctx->ev[4], ctx->ev[5]));
  }
  return CG1_OK;
}

static cg1h::fe fe_from_words12(const uint32_t w[12]) {     // already canonical and in the host's Montgomery form
  cg1h::fe r;
  for (int i = 0; i < 6; ++i) r.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
  return r;
}
static cg1h::jac jac_from_words(const PointWords& p) {
  if (p.inf) return cg1h::jac_identity();
  return cg1h::jac_from_xyzz(fe_from_words12(p.w[0]), fe_from_words12(p.w[1]), fe_from_words12(p.w[2]), fe_from_words12(p.w[3]));
}

}  // namespace cg1
