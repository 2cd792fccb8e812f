// MI355X (gfx950) Pippenger MSM over BLS12-381 G1 -- the per-device pipeline context, the launch chains of the
// two regimes and the C ABI.  The kernels live in kernels_*.h (one translation unit, included below).
//
// Replaces the hot loop of the reference's compute_MSM
//   (/root/reference/curdleproofs/curdleproofs/msm_accumulator.py:6-12:  current += base * scalar)
// with a signed-digit windowed-bucket method laid out for CDNA4.  Regime A (one large MSM), in launch order:
//
//   k_prepare_points   96 B affine (std form) -> 128 B records (x,y as 14x28-bit Montgomery limbs): one aligned
//                      cache line per point, so the bucket gather touches exactly one line     [kernels_prepare_digits.h]
//   k_digits           c-bit signed digits of this rank's windows, window-major u16
//   k_part_count/scatter + k_uscan*, k_bin_sort
//                      two-level LDS partition sort by (window, bucket); no global atomics     [kernels_sort.h]
//                      (k_hist/k_scatter: global-atomic counting sort, only for n > 2^23)
//   k_scan1/2/3, k_chunk_desc, k_len_scan, k_order
//                      bucket offsets; chunks of <= L entries (L = max(8, entries/2^18), <= 4096 chunks per bucket);
//                      chunks ordered by descending length so a wave's lanes finish together
//   k_accumulate  ***  the dominant kernel: one lane per chunk, XYZZ mixed adds over gathered points [kernels_accumulate.h]
//   k_heavy_combine, k_bucket_fold
//                      re-join buckets that were cut into several chunks (skewed scalars, window-sharded ranks)
//   k_rowcol, k_small_tree
//                      2-D bucket reduction: row sums / column sums, then 1 + hb + lb masked sums per window,
//                      exported as canonical XYZZ words                                        [kernels_reduce.h]
//   host tail          one Horner over global bit positions (255 doublings + 256 additions, host_g1.cpp): one GPU
//                      lane needs ~20-30 us per dependent EC operation, a host core ~0.5 us, so the strictly serial
//                      tail belongs on the host.
// Regime B (cg1_msm_batched*): k_group_count/scatter (LDS counting sort per (msm, window)), the same chunking +
// k_accumulate, k_seg_reduce, k_group_reduce, k_msm_horner.
//
// No MFMA: this is carry-propagating big-integer arithmetic.  The bound is the VALU integer-multiply rate
// (v_mad_u64_u32), see fp28.h; bench.py reports both that and the HBM roofline the task sheet asks for.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include "g1_xyzz.h"
#include "g1_quad.h"
#include "host_g1.h"
#include "lazy_host.h"
#include "pool.h"
#include "../../include/curdle_g1.h"

namespace cg1 {

int pick_window(size_t n);

#include "kernels_records.h"
#include "kernels_prepare_digits.h"
#include "kernels_sort.h"
#include "kernels_accumulate.h"
#include "kernels_reduce.h"
#include "kernels_small.h"
#include "kernels_batch.h"
#include "fp_row.h"
}  // namespace cg1
#include "kernels_rows.h"
#include "kernels_merlin.h"
#include "kernels_frontend.h"
#include "kernels_opening.h"
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
    if (!has && done) { armed = true; cv.notify_all(); }
  }
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
  int horner_row = 1;                   // "horner_row": regime B's device Horner with one wave per MSM, one limb per lane (A/B switch; 0: one quad per MSM)
  int batch_mul_row = 1;                // "batch_mul_row": deferred map / fold batches of 96 .. 4096 results on k_batch_mul_row (A/B switch; 0: pool / k_batch_mul)
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
  uint8_t* h_lin = nullptr; size_t cap_h_lin = 0;   // page-locked gather buffer of cg1_lincomb_batch (terms' points | scalars)
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
  if (c->h_lin) { (void)hipHostFree(c->h_lin); c->h_lin = nullptr; c->cap_h_lin = 0; }
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

// c > 0: uniform windows of width c (nwin = 255 / c + 1).  c < 0: a BALANCED plan with cmax = -c: the 256 bit positions
// are cut into nwin = ceil(256 / cmax) windows of width cmax (the low ones) or cmax - 1, so the top window keeps
// >= cmax - 2 scalar bits and the recoding carry never leaves it (scalars are < 2^255).
static WinPlan make_plan(int c) {
  WinPlan pl;
  if (c > 0) { pl.cmax = c; pl.nwin = 255 / c + 1; pl.n_hi = pl.nwin; }
  else { const int cm = -c, nw = (256 + cm - 1) / cm; pl.cmax = cm; pl.nwin = nw; pl.n_hi = 256 - nw * (cm - 1); }
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
  if (resident) HIPCHK(hipMemsetAsync(bad_flag, 0, 16, st));
  else launch_prepare(st, src, ctx->d_pts, ctx->d_flags, n32, bad_flag);
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
    hipLaunchKernelGGL(k_digits, dim3(gn), dim3(256), 0, st, (const uint32_t*)d_scalars32, flags, ctx->d_digits, n32, plan, rank, world, bad_flag);
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
    hipLaunchKernelGGL(k_slice_plan, dim3(1), dim3(256), 0, st, ctx->d_blockcnt, nbt, nslices, ctx->d_slice_base, ctx->d_bigflag, ctx->big_bins);
    hipLaunchKernelGGL(k_bin_sort, dim3(nbt), dim3(256), 0, st, ctx->d_part, ctx->d_blockcnt, ctx->d_hist, ctx->d_sorted, nbt, nslices, sub_bits, ctx->stage_sort, ctx->d_bigflag);
    if (ctx->big_bins && n32 > BIN_STAGE) {                        // a bin cannot exceed n entries
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
  if (small_quad)
    hipLaunchKernelGGL(k_bucket_fold_quad, dim3((uint32_t)((nb_total * 4 + 255) / 256)), dim3(256), 0, st, ctx->d_choff, ctx->d_sums, ctx->d_combined, (uint32_t)nb_total, ctx->d_any_multi);
  else if (ctx->fold_pass)
    hipLaunchKernelGGL(k_bucket_fold, dim3((uint32_t)((nb_total + 255) / 256)), dim3(256), 0, st, ctx->d_choff, ctx->d_sums, ctx->d_combined, (uint32_t)nb_total, ctx->d_any_multi);
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
      hipLaunchKernelGGL(k_rowcol_quad, dim3((rc_waves + 3u) / 4u), dim3(256), 0, st, ctx->d_choff, ctx->d_sums,
                         rowsum, colsum, (uint32_t)nlw, hb2, lb2, lgq);
    }
    else
      hipLaunchKernelGGL(k_rowcol, dim3(nrow_blocks + ncol_blocks), dim3(256), 0, st, ctx->d_choff, ctx->d_sums, ctx->d_combined,
                         rowsum, colsum, (uint32_t)nlw, hb2, lb2, nrow_blocks, ctx->quad);
    if (profile >= 2) HIPCHK(hipEventRecord(ctx->ev[6], st));
    // 4 lanes per element of the longer of the two sums (2^hb rows, 2^lb columns), at most 512 threads: no idle quads in the block
    const uint32_t tree_threads = std::min<uint32_t>(512u, std::max<uint32_t>(64u, 4u << std::max(hb2, lb2)) >> (ctx->tree_shift >= 0 ? ctx->tree_shift : (ctx->tree_half ? 1 : 0)));
    if (ctx->quad) hipLaunchKernelGGL(k_small_tree_quad, dim3(nitems, nlw), dim3(std::max<uint32_t>(64u, tree_threads)), 0, st, rowsum, colsum, ctx->d_out, hb2, lb2);
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
  if (zc) {
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
  if (pd.zero_copy && !ctx->blocking_sync && pd.profile < 2) {
    if (pd.arm_helpers) for (int j = 0; j < 3 && j + 1 < ctx->horner_threads; ++j) ctx->helper[j].arm();
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
    auto put = [&](int e, const PointWords* p) { if (!p->inf) { items.emplace_back(e, p); if (e > e_top) e_top = e; } };
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
  const int nth = (!ctx->host_split || e_top < 96) ? 1 : (ctx->horner_threads >= 4 && e_top >= 192 ? 4 : 2);
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
    run_part(nth - 1);
    acc = part[nth - 1];
    for (int j = 0; j + 1 < nth; ++j) { ctx->helper[j].wait(); acc = cg1h::jac_add(acc, part[j]); }
  } else {
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
static int msm_enqueue_small(Ctx* ctx, const PtSrc& src, const void* d_scalars32, size_t n, int c, uint32_t M = 1, const uint32_t* d_offs = nullptr, size_t max_n = 0) {
  ctx->pend.active = false;
  HIPCHK(hipSetDevice(ctx->device));
  const WinPlan plan = make_plan(c);
  const uint32_t nwin = (uint32_t)plan.nwin, bb = (uint32_t)c - 1u, lb2 = (bb + 1u) / 2u, hb2 = bb - lb2, nitems = 1u + hb2 + lb2;
  if (M == 1) max_n = n;
  const uint32_t S = (uint32_t)((max_n + SM_SLICE - 1) / SM_SLICE);
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
  if (ctx->small_msm && n <= SM_MAX_N && world == 1 && (c == 0 || (c >= 4 && c <= 9 && c != 5)) &&
      (size_t)(255 / (c ? c : pick_small_c(n)) + 1) * ((n + SM_SLICE - 1) / SM_SLICE) <= SM_ONE_ROUND) {
    if (c == 0) c = pick_small_c(n);
    ctx->pend_c = c;
    return msm_enqueue_small(ctx, src, d_scalars32, n, c);
  }
  if (c == 0) c = pick_plan_c(n, ctx->auto_plan);
  const int cabs = c < 0 ? -c : c;
  if (cabs < 4 || cabs > 16) { snprintf(ctx->err, sizeof ctx->err, "window width %d out of range [4,16]", c); return CG1_ERR_ARG; }
  const WinPlan plan = make_plan(c);
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
// async_small (may be NULL): when the call fits ONE k_msm_small launch it is only ENQUEUED and *async_small set; the caller does other
// work and collects the results with msm_batched_small_end.  Calls that take the regime-B chain complete before returning.
static int msm_batched_small_end(Ctx* ctx, size_t M, std::vector<cg1h::jac>& results) {
  if (M == 1) { cg1h::jac r; int rc = msm_finish(ctx, r); ctx->last_acc_launches = 0; if (rc == CG1_OK) results[0] = r; return rc; }
  return msm_small_batched_finish(ctx, (uint32_t)M, results);
}
int msm_batched_device(Ctx* ctx, const void* d_points96, const void* d_scalars32, const uint32_t* h_offsets, size_t M,
                       int c, std::vector<cg1h::jac>& results, bool* async_small = nullptr) {
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
    const int cs = c > 0 ? c : pick_small_c(max_n);
    const size_t groups = (size_t)M * (size_t)(255 / cs + 1) * ((max_n + SM_SLICE - 1) / SM_SLICE);
    if (ctx->small_msm && M <= SM_MAX_MSMS && max_n <= SM_MAX_N && groups <= SM_MAX_GROUPS && cs >= 4 && cs <= 9 && cs != 5) {
      HIPCHK(hipSetDevice(ctx->device));
      if ((M + 1) > ctx->cap_boffs) {
        if (ctx->d_boffs) (void)hipFree(ctx->d_boffs);
        ctx->d_boffs = nullptr; ctx->cap_boffs = 0;
        HIPCHK(hipMalloc(&ctx->d_boffs, (M + 1) * 4));
        ctx->cap_boffs = M + 1;
      }
      HIPCHK(hipMemcpyAsync(ctx->d_boffs, h_offsets, (M + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
      int rc = msm_enqueue_small(ctx, PtSrc(d_points96), d_scalars32, N, cs, (uint32_t)M, ctx->d_boffs, max_n);
      if (rc) return rc;
      if (async_small) { *async_small = true; return CG1_OK; }
      return msm_batched_small_end(ctx, M, results);
    }
  }
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
    HIPCHK(hipMalloc(&ctx->d_bout, (M + 1) * sizeof(PointWords)));         // + one record: the input-validation flag word
    HIPCHK(hipHostMalloc(&ctx->h_bout, (M + 1) * sizeof(PointWords)));
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
  if (ctx->profile >= 2) HIPCHK(hipEventRecord(ctx->ev[0], st));
  uint32_t* bad_flag = reinterpret_cast<uint32_t*>(ctx->d_bout + M);
  hipLaunchKernelGGL(k_prepare_points, dim3(gn), dim3(256), 0, st, (const uint32_t*)d_points96, ctx->d_pts, ctx->d_flags, N32, bad_flag);
  if (ctx->profile >= 2) HIPCHK(hipEventRecord(ctx->ev[1], st));
  hipLaunchKernelGGL(k_digits, dim3(gn), dim3(256), 0, st, (const uint32_t*)d_scalars32, ctx->d_flags, ctx->d_digits, N32, make_plan(c), 0, 1, bad_flag);
  hipLaunchKernelGGL(k_group_count, dim3((uint32_t)M, nwin), dim3(256), 0, st, ctx->d_digits, ctx->d_boffs, ctx->d_hist, N32, NB, nwin);
  if (ctx->profile >= 2) HIPCHK(hipEventRecord(ctx->ev[2], st));
  const uint32_t nblk = (uint32_t)((nb_total + SCAN_ITEMS - 1) / SCAN_ITEMS);
  hipLaunchKernelGGL(k_scan1, dim3(nblk), dim3(256), 0, st, ctx->d_hist, ctx->d_off, ctx->d_choff, ctx->d_blocktot, (uint32_t)nb_total, L0);
  hipLaunchKernelGGL(k_scan2, dim3(1), dim3(256), 0, st, ctx->d_blocktot, nblk, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
  hipLaunchKernelGGL(k_scan3, dim3(nblk), dim3(256), 0, st, ctx->d_blocktot, ctx->d_off, ctx->d_choff, (uint32_t)nb_total);
  hipLaunchKernelGGL(k_group_scatter, dim3((uint32_t)M, nwin), dim3(256), 0, st, ctx->d_digits, ctx->d_boffs, ctx->d_off, ctx->d_sorted, N32, NB, nwin);
  if (ctx->profile >= 2) HIPCHK(hipEventRecord(ctx->ev[3], st));
  HIPCHK(hipMemsetAsync(ctx->d_zblock, 0, zblock_clear_bytes(ctx), st));
  hipLaunchKernelGGL(k_chunk_desc, dim3((uint32_t)std::min<size_t>(CHUNK_DESC_BLOCKS, (nb_total + 255) / 256)), dim3(256), 0, st, ctx->d_off, ctx->d_choff, ctx->d_desc, ctx->d_lenhist, ctx->d_heavy, (uint32_t)ctx->cap_heavy, (uint32_t)nb_total, L0, ctx->d_any_multi);
  const size_t max_chunks = nb_total + (N * (size_t)nwin) / L0 + 1;
  const uint32_t gchunks = (uint32_t)((max_chunks + 255) / 256);
  hipLaunchKernelGGL(k_len_scan, dim3(1), dim3(256), 0, st, ctx->d_lenhist, ctx->d_lenhist + LEN_BINS, ctx->d_off + nb_total, ctx->d_choff + nb_total, bad_flag);
  hipLaunchKernelGGL(k_order, dim3((gchunks + ORDER_PER - 1) / ORDER_PER), dim3(256), 0, st, ctx->d_desc, ctx->d_choff + nb_total, ctx->d_lenhist + LEN_BINS, ctx->d_order);
  if (ctx->profile >= 1) HIPCHK(hipEventRecord(ctx->ev[4], st));
  hipLaunchKernelGGL(k_accumulate, dim3(gchunks), dim3(256), 0, st, ctx->d_desc, ctx->d_choff + nb_total, ctx->d_order, ctx->d_sorted, ctx->d_pts, ctx->d_sums);
  if (ctx->profile >= 1) HIPCHK(hipEventRecord(ctx->ev[5], st));
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
  auto h1 = std::chrono::steady_clock::now();
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

}  // namespace cg1

// ================================================================== C ABI (include/curdle_g1.h)
using cg1::Ctx;
struct cg1_ctx : public cg1::Ctx {};

namespace cg1 {
static Ctx* child_of(Ctx* ctx) { return static_cast<Ctx*>(ctx->child); }

// One call as two launch chains: this context takes the HIGH half of the plan's windows (and prepares the points), its child the
// LOW half on its own stream.  The child starts once the prepared records exist, and its k_accumulate waits for this context's to
// finish: the two dominant launches run back to back, everything around them overlaps with one of them.
static int msm_begin_split(Ctx* ctx, const PtSrc& src, const void* d_scalars32, size_t n, const WinPlan& plan, int rank, int world) {
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->child) {
    cg1_ctx* made = ctx->cu_mask.empty() ? cg1_ctx_create(ctx->device) : cg1_ctx_create_cu_mask(ctx->device, ctx->cu_mask.data(), ctx->cu_mask.size());
    if (!made) { snprintf(ctx->err, sizeof ctx->err, "could not create the second launch chain's context"); return CG1_ERR_HIP; }
    made->split = 0;
    ctx->child = made;
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_acc, hipEventDisableTiming));
  }
  Ctx* ch = child_of(ctx);
  ch->profile = ctx->profile; ch->L0 = ctx->L0; ch->seg_m = ctx->seg_m; ch->quad = ctx->quad; ch->reduce_2d = ctx->reduce_2d;
  ch->rowcol_quad = ctx->rowcol_quad; ch->rowcol_quad_max = ctx->rowcol_quad_max; ch->fold_pass = ctx->fold_pass; ch->tree_half = ctx->tree_half; ch->tree_shift = ctx->tree_shift; ch->rowcol_lgq = ctx->rowcol_lgq; ch->sort_sub_bits = ctx->sort_sub_bits;
  ch->scan_one = ctx->scan_one; ch->zero_copy = ctx->zero_copy; ch->horner_threads = ctx->horner_threads; ch->host_split = ctx->host_split;
  ch->blocking_sync = ctx->blocking_sync; ch->stage_sort = ctx->stage_sort; ch->use_partition_sort = ctx->use_partition_sort; ch->big_bins = ctx->big_bins;
  const int n_own = win_count(plan.nwin, rank, world), n_lo = n_own / 2, n_hi = n_own - n_lo;      // this rank's windows: the upper ones here, the lower ones on the child
  ChainHooks hi;
  hi.after_prepare = ctx->ev_prep; hi.after_accumulate = ctx->ev_acc;
  int rc = msm_enqueue(ctx, src, d_scalars32, n, plan, rank, win_sel(world, n_lo, n_hi), hi);
  if (rc) return rc;
  PtSrc shared;
  shared.kind = PtSrc::PREPARED;
  shared.p = src.kind == PtSrc::PREPARED ? src.p : ctx->d_pts;
  shared.flags = src.kind == PtSrc::PREPARED ? src.flags : ctx->d_flags;
  ChainHooks lo;
  lo.before_start = ctx->ev_prep; lo.before_accumulate = ctx->ev_acc;
  rc = msm_enqueue(ch, shared, d_scalars32, n, plan, rank, win_sel(world, 0, n_lo), lo);
  if (rc) { snprintf(ctx->err, sizeof ctx->err, "%s", ch->err); cg1h::jac dummy; (void)msm_finish(ctx, dummy); return rc; }
  ctx->pend_split = true;
  return CG1_OK;
}
}  // namespace cg1

namespace {
struct DevBuf {                       // frees on every exit path of the host-pointer convenience entry points
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
};
}  // namespace

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
int cg1_batch_from_affine96(uint8_t* out_blobs, const uint8_t* in96, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    int rc = cg1_from_affine96(out_blobs + CG1_POINT_BYTES * i, in96 + 96 * i, 0);
    if (rc) return rc;
  }
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

// cu_mask (n_words x 32 bits, bit i = compute unit i; NULL = every CU): the context's compute and side streams only run on those CUs
// (hipExtStreamCreateWithCUMask).  The verifier with its front-end on the device gives its latency-bound front-end launches a few CUs
// of their own and keeps the throughput kernels (decompression, MSM) off them.
cg1_ctx* cg1_ctx_create_cu_mask(int device, const uint32_t* cu_mask, size_t n_words) {
  int n = cg1_device_count();
  if (device < 0 || device >= n) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  cg1_ctx* ctx = new cg1_ctx();
  ctx->device = device;
  hipError_t e;
  if (cu_mask && n_words) {
    e = hipExtStreamCreateWithCUMask(&ctx->stream, (uint32_t)n_words, cu_mask);
    ctx->cu_mask.assign(cu_mask, cu_mask + n_words);
  } else {
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  }
  if (e != hipSuccess) { delete ctx; return nullptr; }
  if (hipEventCreateWithFlags(&ctx->copy_ev, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&ctx->sync_ev, hipEventDisableTiming | hipEventBlockingSync) != hipSuccess) { delete ctx; return nullptr; }
  for (int i = 0; i <= CG1_NPHASE; ++i) if (hipEventCreate(&ctx->ev[i]) != hipSuccess) { delete ctx; return nullptr; }
  if (hipHostMalloc((void**)&ctx->h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess || hipHostGetDevicePointer((void**)&ctx->h_flag_dev, ctx->h_flag, 0) != hipSuccess) { delete ctx; return nullptr; }
  *ctx->h_flag = 0;
  return ctx;
}
cg1_ctx* cg1_ctx_create(int device) { return cg1_ctx_create_cu_mask(device, nullptr, 0); }
void cg1_ctx_destroy(cg1_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->child) { cg1_ctx_destroy(ctx->child); ctx->child = nullptr; }
  if (ctx->ev_prep) (void)hipEventDestroy(ctx->ev_prep);
  if (ctx->ev_acc) (void)hipEventDestroy(ctx->ev_acc);
  cg1::free_bufs(ctx);
  if (ctx->d_merlin_rows) (void)hipFree(ctx->d_merlin_rows);
  if (ctx->d_opening) (void)hipFree(ctx->d_opening);
  if (ctx->d_small_partial) (void)hipFree(ctx->d_small_partial);
  if (ctx->d_small_ctr) (void)hipFree(ctx->d_small_ctr);
  if (ctx->d_small_pts) (void)hipFree(ctx->d_small_pts);
  if (ctx->d_small_flags) (void)hipFree(ctx->d_small_flags);
  if (ctx->h_small_out) (void)hipHostFree(ctx->h_small_out);
  if (ctx->d_stage_pts) (void)hipFree(ctx->d_stage_pts);
  if (ctx->d_stage_sc) (void)hipFree(ctx->d_stage_sc);
  if (ctx->h_flag) (void)hipHostFree(ctx->h_flag);
  for (int i = 0; i <= CG1_NPHASE; ++i) (void)hipEventDestroy(ctx->ev[i]);
  (void)hipStreamDestroy(ctx->stream);
  if (ctx->copy_ev) (void)hipEventDestroy(ctx->copy_ev);
  if (ctx->sync_ev) (void)hipEventDestroy(ctx->sync_ev);
  for (int i = 0; i < 2; ++i) if (ctx->tm_ev[i]) (void)hipEventDestroy(ctx->tm_ev[i]);
  if (ctx->copy_stream.load()) (void)hipStreamDestroy(ctx->copy_stream.load());
  if (ctx->side_ev) (void)hipEventDestroy(ctx->side_ev);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
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
// Asynchronous H2D on the context's copy stream (src must be page-locked for the copy to overlap kernels), and the
// fence that orders everything queued on the copy stream so far before whatever is launched next on the compute
// stream.  Neither blocks the host.  cg1_h2d_async touches only the copy stream: it may be called from a second
// thread while another thread runs kernels on this context.
int cg1_h2d_async(cg1_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return CG1_ERR_HIP;
  if (bytes == 0) return CG1_OK;
  HIPCHK(hipSetDevice(ctx->device));
  std::call_once(ctx->copy_once, [ctx]() { hipStream_t s = nullptr; if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess) ctx->copy_stream.store(s); });
  hipStream_t cs = ctx->copy_stream.load();
  if (!cs) { snprintf(ctx->err, sizeof ctx->err, "could not create the copy stream"); return CG1_ERR_HIP; }
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs));
  return CG1_OK;
}
// wait for the context's compute stream only (cg1_ctx_sync waits for the whole device, other contexts included)
int cg1_stream_sync(cg1_ctx* ctx) {
  if (!ctx) return CG1_ERR_HIP;
  HIPCHK(hipSetDevice(ctx->device));
  return cg1::wait_stream(ctx);
}
int cg1_copy_fence(cg1_ctx* ctx) {
  if (!ctx) return CG1_ERR_HIP;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t cs = ctx->copy_stream.load();
  if (!cs) return CG1_OK;                                  // nothing was ever queued on it
  HIPCHK(hipEventRecord(ctx->copy_ev, cs));
  HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->copy_ev, 0));
  return CG1_OK;
}
// page-locked host memory: H2D/D2H copies from it run at full PCIe rate (pageable memory is staged by the runtime)
void* cg1_host_alloc(cg1_ctx* ctx, size_t bytes) {
  if (!ctx) return nullptr;
  void* p = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1) != hipSuccess) return nullptr;
  return p;
}
void cg1_host_free(cg1_ctx* ctx, void* p) {
  if (!ctx || !p) return;
  (void)hipSetDevice(ctx->device);
  (void)hipHostFree(p);
}
// `rows` records of `width` bytes, `src_pitch` apart on the device, packed `dst_pitch` apart on the host
int cg1_d2h_2d(cg1_ctx* ctx, void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t rows) {
  if (!ctx) return CG1_ERR_HIP;
  if (rows == 0 || width == 0) return CG1_OK;
  if (width > dst_pitch || width > src_pitch) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpy2D(dst, dst_pitch, src, src_pitch, width, rows, hipMemcpyDeviceToHost));
  return CG1_OK;
}
int cg1_ctx_sync(cg1_ctx* ctx) {
  if (!ctx) return CG1_ERR_HIP;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());
  return CG1_OK;
}
int cg1_merlin_last_passes(const cg1_ctx* ctx) { return ctx ? (int)ctx->merlin_passes : -1; }
int cg1_merlin_last_kernel(const cg1_ctx* ctx) { return ctx ? ctx->merlin_last_kernel : -1; }
int cg1_ctx_device(const cg1_ctx* ctx) { return ctx ? ctx->device : -1; }
void* cg1_ctx_stream(cg1_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int cg1_ctx_set_param(cg1_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return CG1_ERR_ARG;
  if (!strcmp(name, "chunk_rule")) { ctx->chunk_rule = value != 0; return CG1_OK; }
  if (!strcmp(name, "chunk_len")) { if (value < 1 || value > 65536) return CG1_ERR_ARG; ctx->L0 = (uint32_t)value; cg1::free_bufs(ctx); return CG1_OK; }
  if (!strcmp(name, "stage_sort")) { ctx->stage_sort = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "quad")) { ctx->quad = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "horner_row")) { ctx->horner_row = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "batch_mul_row")) { ctx->batch_mul_row = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "small_msm")) { ctx->small_msm = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "split")) { ctx->split = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "split_min_log2n")) { if (value < 10 || value > 31) return CG1_ERR_ARG; ctx->split_min_n = (size_t)1 << value; return CG1_OK; }
  if (!strcmp(name, "reduce_2d")) { ctx->reduce_2d = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "partition_sort")) { ctx->use_partition_sort = value ? 1 : 0; cg1::free_bufs(ctx); return CG1_OK; }
  if (!strcmp(name, "blocking_sync")) { ctx->blocking_sync = value != 0; return CG1_OK; }
  if (!strcmp(name, "big_bins")) { ctx->big_bins = value != 0; return CG1_OK; }
  if (!strcmp(name, "host_split")) { ctx->host_split = value != 0; return CG1_OK; }
  if (!strcmp(name, "arm_helpers")) { ctx->arm_helpers = value != 0; return CG1_OK; }
  if (!strcmp(name, "horner_threads")) { if (value != 1 && value != 2 && value != 4) return CG1_ERR_ARG; ctx->horner_threads = value; return CG1_OK; }
  if (!strcmp(name, "zero_copy")) { ctx->zero_copy = value != 0; return CG1_OK; }
  if (!strcmp(name, "auto_plan")) { ctx->auto_plan = value != 0; return CG1_OK; }
  if (!strcmp(name, "rowcol_quad")) { ctx->rowcol_quad = value != 0; return CG1_OK; }
  if (!strcmp(name, "rowcol_quad_max")) { if (value < 0) return CG1_ERR_ARG; ctx->rowcol_quad_max = value; return CG1_OK; }
  if (!strcmp(name, "fold_pass")) { ctx->fold_pass = value != 0; return CG1_OK; }
  if (!strcmp(name, "scan_one")) { ctx->scan_one = value != 0; return CG1_OK; }
  if (!strcmp(name, "batched_host_horner_max")) { if (value < 0) return CG1_ERR_ARG; ctx->batched_host_horner_max = value; return CG1_OK; }
  if (!strcmp(name, "tree_shift")) { if (value < -1 || value > 4) return CG1_ERR_ARG; ctx->tree_shift = value; return CG1_OK; }
  if (!strcmp(name, "rowcol_lgq")) { if (value != 0 && (value < 2 || value > 4)) return CG1_ERR_ARG; ctx->rowcol_lgq = value; return CG1_OK; }
  if (!strcmp(name, "sort_sub_bits")) { if (value != 0 && (value < 4 || value > 8)) return CG1_ERR_ARG; ctx->sort_sub_bits = value; return CG1_OK; }
  if (!strcmp(name, "batch_mul_host_max")) { if (value < -1) return CG1_ERR_ARG; ctx->batch_mul_host_max = value; return CG1_OK; }
  if (!strcmp(name, "batch_mul_quad_max")) { if (value < 0) return CG1_ERR_ARG; ctx->batch_mul_quad_max = value; return CG1_OK; }
  if (!strcmp(name, "merlin_sync")) { ctx->merlin_sync = value != 0; return CG1_OK; }
  if (!strcmp(name, "fe_timed")) { ctx->fe_timed = value != 0; return CG1_OK; }
  if (!strcmp(name, "fe_rows")) { ctx->fe_rows = value != 0; return CG1_OK; }
  if (!strcmp(name, "fe_prio")) { if (value < 0 || value > 3) return CG1_ERR_ARG; ctx->fe_prio = value; return CG1_OK; }
  if (!strcmp(name, "decompress_waves")) { if (value != 2 && value != 3) return CG1_ERR_ARG; ctx->decompress_waves = value; return CG1_OK; }
  if (!strcmp(name, "merlin_rows")) { ctx->merlin_rows = value != 0; return CG1_OK; }
  if (!strcmp(name, "merlin_lanes")) { if (value < 1 || value > 64) return CG1_ERR_ARG; ctx->merlin_lanes = value; return CG1_OK; }
  if (!strcmp(name, "tree_half")) { ctx->tree_half = value != 0; return CG1_OK; }
  if (!strcmp(name, "wave_agg")) {
    int v = value ? 1 : 0;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(cg1::g_wave_agg), &v, sizeof v));
    return CG1_OK;
  }
  if (!strcmp(name, "profile")) { if (value < 0 || value > 2) return CG1_ERR_ARG; ctx->profile = value; return CG1_OK; }
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

// The same call in two halves: _begin enqueues this context's whole launch chain and returns at once, _end waits for it,
// runs the host tail and delivers the point.  With two contexts on one GPU, begin the next MSM before ending this one.
int cg1_msm_device_begin(cg1_ctx* ctx, const void* d_points, const void* d_scalars, size_t n, int window_c, int shard_rank, int shard_world) {
  if (!ctx) return CG1_ERR_HIP;
  return cg1::msm_begin(ctx, d_points, d_scalars, n, window_c, shard_rank, shard_world);
}
int cg1_msm_device_end(cg1_ctx* ctx, uint8_t* out) {
  if (!ctx) return CG1_ERR_HIP;
  cg1h::jac r;
  int rc = cg1::msm_end(ctx, r);
  if (rc == CG1_OK) blob_out(out, r);
  return rc;
}

// One MSM over several GPUs of THIS process: context i owns point shard i on its own device.  Every launch chain is enqueued
// before any is waited for, so the devices work concurrently; the partials are added in context order.
int cg1_msm_multi_device(cg1_ctx* const* ctxs, size_t n_ctx, const void* const* d_points, const void* const* d_scalars, const size_t* n,
                         int window_c, uint8_t* out) {
  if (!ctxs || !n_ctx || !d_points || !d_scalars || !n || !out) return CG1_ERR_ARG;
  for (size_t i = 0; i < n_ctx; ++i) {
    if (!ctxs[i]) return CG1_ERR_HIP;
    for (size_t j = 0; j < i; ++j) if (ctxs[j] == ctxs[i]) return CG1_ERR_ARG;        // a context takes one call at a time
  }
  int rc = CG1_OK;
  size_t begun = 0;
  for (; begun < n_ctx && rc == CG1_OK; ++begun)
    rc = cg1::msm_begin(ctxs[begun], d_points[begun], d_scalars[begun], n[begun], window_c, 0, 1);
  cg1h::jac acc = cg1h::jac_identity();
  for (size_t i = 0; i < begun; ++i) {                     // drain every context that was begun, also after a failure
    cg1h::jac part;
    int r2 = cg1::msm_end(ctxs[i], part);
    if (rc == CG1_OK) rc = r2;
    if (r2 == CG1_OK) acc = cg1h::jac_add(acc, part);
  }
  if (rc == CG1_OK) blob_out(out, acc);
  return rc;
}

// device staging for the host-pointer entry points (grown geometrically, kept by the context)
static int ensure_stage(cg1_ctx* ctx, size_t pts_bytes, size_t sc_bytes) {
  if (pts_bytes > ctx->cap_stage_pts) {
    if (ctx->d_stage_pts) (void)hipFree(ctx->d_stage_pts);
    ctx->d_stage_pts = nullptr; ctx->cap_stage_pts = 0;
    const size_t want = pts_bytes + pts_bytes / 4 + 256;
    HIPCHK(hipMalloc(&ctx->d_stage_pts, want));
    ctx->cap_stage_pts = want;
  }
  if (sc_bytes > ctx->cap_stage_sc) {
    if (ctx->d_stage_sc) (void)hipFree(ctx->d_stage_sc);
    ctx->d_stage_sc = nullptr; ctx->cap_stage_sc = 0;
    const size_t want = sc_bytes + sc_bytes / 4 + 256;
    HIPCHK(hipMalloc(&ctx->d_stage_sc, want));
    ctx->cap_stage_sc = want;
  }
  return CG1_OK;
}

int cg1_msm(cg1_ctx* ctx, const uint8_t* points, const uint8_t* scalars, size_t n, uint8_t* out) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) { blob_out(out, cg1h::jac_identity()); return CG1_OK; }
  HIPCHK(hipSetDevice(ctx->device));
  { int src = ensure_stage(ctx, n * 96, n * 32); if (src) return src; }
  HIPCHK(hipMemcpyAsync(ctx->d_stage_pts, points, n * 96, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_stage_sc, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
  return cg1_msm_device(ctx, ctx->d_stage_pts, ctx->d_stage_sc, n, 0, 0, 1, out);
}

// compute_MSM over the point blobs G1Point objects hold (host memory; page-locked staging copies at full PCIe rate): uploaded as
// they are, normalised on the device (k_prepare_blobs).  all_normalised != 0: the caller knows every Z is 0 or 1.
int cg1_msm_blobs(cg1_ctx* ctx, const uint8_t* blobs144, const uint8_t* scalars32, size_t n, int all_normalised, uint8_t* out) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) { blob_out(out, cg1h::jac_identity()); return CG1_OK; }
  if (!blobs144 || !scalars32) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  { int src = ensure_stage(ctx, n * CG1_POINT_BYTES, n * 32); if (src) return src; }
  HIPCHK(hipMemcpyAsync(ctx->d_stage_pts, blobs144, n * CG1_POINT_BYTES, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_stage_sc, scalars32, n * 32, hipMemcpyHostToDevice, ctx->stream));
  cg1::PtSrc src;
  src.kind = cg1::PtSrc::BLOBS; src.p = ctx->d_stage_pts; src.normalised = all_normalised != 0;
  cg1h::jac r;
  int rc = cg1::msm_device(ctx, src, ctx->d_stage_sc, n, 0, 0, 1, r);
  if (rc == CG1_OK) blob_out(out, r);
  return rc;
}
// The two halves of cg1_msm_blobs for a caller that uploads in slices while it is still gathering (msm_accumulator.compute_MSM over 2^20
// objects: each 64 K-element slice is copied by cg1_h2d_async while the next one is packed): cg1_stage_reserve hands out the context's
// device staging (valid until the next call that stages more), cg1_msm_blobs_device runs the MSM over blobs already there.
int cg1_stage_reserve(cg1_ctx* ctx, size_t pts_bytes, size_t sc_bytes, void** d_pts, void** d_sc) {
  if (!ctx) return CG1_ERR_HIP;
  HIPCHK(hipSetDevice(ctx->device));
  { int src = ensure_stage(ctx, pts_bytes, sc_bytes); if (src) return src; }
  if (d_pts) *d_pts = ctx->d_stage_pts;
  if (d_sc) *d_sc = ctx->d_stage_sc;
  return CG1_OK;
}
int cg1_msm_blobs_device(cg1_ctx* ctx, const void* d_blobs144, const void* d_scalars32, size_t n, int all_normalised, uint8_t* out) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) { blob_out(out, cg1h::jac_identity()); return CG1_OK; }
  if (!d_blobs144 || !d_scalars32) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  cg1::PtSrc src;
  src.kind = cg1::PtSrc::BLOBS; src.p = d_blobs144; src.normalised = all_normalised != 0;
  cg1h::jac r;
  int rc = cg1::msm_device(ctx, src, d_scalars32, n, 0, 0, 1, r);
  if (rc == CG1_OK) blob_out(out, r);
  return rc;
}
}  // extern "C"

// A vector of points kept on the device in the accumulation kernels' own record format (128 B per point + a flag byte): made once
// from the host objects' blobs, used by any number of MSMs (crs.vec_G / vec_H across a prover's dozens of compute_MSM calls).
struct cg1_vec {
  int device = 0;
  size_t n = 0;
  cg1::PreparedPoint* d_pts = nullptr;
  uint8_t* d_flags = nullptr;
};

extern "C" {
cg1_vec* cg1_vec_create(cg1_ctx* ctx, const uint8_t* blobs144, size_t n, int all_normalised) {
  if (!ctx || (!blobs144 && n) || n >= (1ull << 31)) return nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
  cg1_vec* v = new cg1_vec();
  v->device = ctx->device; v->n = n;
  if (hipMalloc(&v->d_pts, (n ? n : 1) * sizeof(cg1::PreparedPoint)) != hipSuccess || hipMalloc(&v->d_flags, n + 16) != hipSuccess) { cg1_vec_destroy(v); return nullptr; }
  if (n == 0) return v;
  if (ensure_stage(ctx, n * CG1_POINT_BYTES, 0) != CG1_OK) { cg1_vec_destroy(v); return nullptr; }
  if (hipMemcpyAsync(ctx->d_stage_pts, blobs144, n * CG1_POINT_BYTES, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { cg1_vec_destroy(v); return nullptr; }
  cg1::PtSrc src;
  src.kind = cg1::PtSrc::BLOBS; src.p = ctx->d_stage_pts; src.normalised = all_normalised != 0;
  cg1::launch_prepare(ctx->stream, src, v->d_pts, v->d_flags, (uint32_t)n, nullptr);
  if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) { cg1_vec_destroy(v); return nullptr; }
  return v;
}
void cg1_vec_destroy(cg1_vec* v) {
  if (!v) return;
  (void)hipSetDevice(v->device);
  if (v->d_pts) (void)hipFree(v->d_pts);
  if (v->d_flags) (void)hipFree(v->d_flags);
  delete v;
}
size_t cg1_vec_len(const cg1_vec* v) { return v ? v->n : 0; }
// sum_{i < n} scalars[i] * vec[first + i]; scalars in host memory
int cg1_msm_vec(cg1_ctx* ctx, const cg1_vec* vec, size_t first, size_t n, const uint8_t* scalars32, uint8_t* out) {
  if (!ctx) return CG1_ERR_HIP;
  if (!vec || first > vec->n || n > vec->n - first || vec->device != ctx->device) return CG1_ERR_ARG;
  if (n == 0) { blob_out(out, cg1h::jac_identity()); return CG1_OK; }
  if (!scalars32) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  { int src = ensure_stage(ctx, 0, n * 32); if (src) return src; }
  HIPCHK(hipMemcpyAsync(ctx->d_stage_sc, scalars32, n * 32, hipMemcpyHostToDevice, ctx->stream));
  cg1::PtSrc src;
  src.kind = cg1::PtSrc::PREPARED; src.p = vec->d_pts + first; src.flags = vec->d_flags + first;
  cg1h::jac r;
  int rc = cg1::msm_device(ctx, src, ctx->d_stage_sc, n, 0, 0, 1, r);
  if (rc == CG1_OK) blob_out(out, r);
  return rc;
}
// Host: n point blobs -> affine96 and / or compressed48 (either may be NULL) with ONE shared inversion -- what
// MSMAccumulator.accumulate_check needs of its bases: the map key (48-byte compression, msm_accumulator.py:54) and the affine form
int cg1_batch_normalize(const uint8_t* blobs, size_t n, uint8_t* out_affine96, uint8_t* out_comp48) {
  if (n && !blobs) return CG1_ERR_ARG;
  std::vector<cg1h::jac> pts(n);
  std::vector<cg1h::fe> xs(n), ys(n);
  std::vector<uint8_t> inf(n);
  for (size_t i = 0; i < n; ++i) pts[i] = blob_in(blobs + CG1_POINT_BYTES * i);
  cg1h::jac_batch_to_affine(pts.data(), n, xs.data(), ys.data(), inf.data());
  for (size_t i = 0; i < n; ++i) {
    if (out_affine96) {
      uint8_t* o = out_affine96 + 96 * i;
      if (inf[i]) memset(o, 0, 96);
      else { cg1h::fe_to_le48(xs[i], o); cg1h::fe_to_le48(ys[i], o + 48); }
    }
    if (out_comp48) cg1h::g1_compress_affine(xs[i], ys[i], inf[i] != 0, out_comp48 + 48 * i);
  }
  return CG1_OK;
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
  { int src = ensure_stage(ctx, n * 96, n * 32); if (src) return src; }
  HIPCHK(hipMemcpyAsync(ctx->d_stage_pts, points, n * 96, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ctx->d_stage_sc, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
  return cg1_msm_batched_device(ctx, ctx->d_stage_pts, ctx->d_stage_sc, offsets, n_msm, 0, out_blobs);
}

// A batch of linear combinations over shared bases -- what a flush of deferred G1Point operators is (py_arkworks_bls12381.py):
// out_j = sum_{t in [offsets[j], offsets[j+1])} scalars[t] * (+/-) bases[term_base[t] & 0x7fffffff]   (bit 31: the negated base).
//
// Two engines.  The host's worker pool evaluates one combination per thread (interleaved width-5 NAF: 255 doublings + ~52 additions per
// term, ~0.25 us each); the GPU evaluates many terms at once but every output ends in a host Horner of 255 dependent doublings, so a
// combination of one to three terms gains nothing from the trip.  path 0 chooses from the BATCH alone (never from the machine: the pool is
// priced at a nominal 8 threads, a k_msm_small launch at 0.25 ms + 20 us per output, the regime-B chain at 1.9 ms -- the round-5
// measurements, profiles/r05_lazy_profile.txt):
//     all on the pool  |  combinations of >= 4 weighted terms on the GPU with the small ones on the pool MEANWHILE (path_used 3)  |  all on the GPU
// path 1 = pool, 2 = GPU (everything gathered into one cg1_msm_batched_device input).  Outputs are normalised: blobs with Z = 1 (or the
// identity), affine96, compressed48 (each may be NULL).
extern "C" void cg1_lincomb_write_outputs(const void* jac_results, size_t n_out, uint8_t* out_blobs144, uint8_t* out_affine96, uint8_t* out_comp48);
}  // extern "C"

constexpr size_t LINCOMB_ROW_MAX = 4096;         // map / fold results k_batch_mul_row takes (one wave each); beyond: k_batch_mul, one lane each
constexpr size_t LINCOMB_ROW_MIN = 96;           // fewer are quicker on the host's pool (~77 us each over its threads) than a ~0.7 ms launch
constexpr size_t LINCOMB_MAX_REGIME_B = 2048;    // independent MSMs cg1_lincomb_batch hands the regime-B chain in one call (r04: 1 024 - 2 048 x 627 terms)

// results[sel[q]] = s * B (+ A) for the selected outputs, each one weighted term and at most one unit term: one k_batch_mul launch
static void negate_affine96_y(uint8_t* rec) {
  uint64_t y[6], any = 0;
  memcpy(y, rec + 48, 48);
  for (int i = 0; i < 6; ++i) any |= y[i];
  if (!any) return;                                      // the identity record stays all-zero
  unsigned __int128 br = 0;
  for (int i = 0; i < 6; ++i) { const unsigned __int128 d = (unsigned __int128)cg1::H_P[i] - y[i] - br; y[i] = (uint64_t)d; br = (d >> 64) & 1; }
  memcpy(rec + 48, y, 48);
}
static int lincomb_shaped_device(cg1_ctx* ctx, const uint8_t* bases_affine96, const uint32_t* offsets, const uint32_t* term_base, const uint8_t* term_scalars32,
                                 const std::vector<uint32_t>& sel, std::vector<cg1h::jac>& results) {
  const size_t m = sel.size();
  if (m == 0) return CG1_OK;
  std::vector<uint8_t> hb(m * 96), hs(m * 32), ha(m * 96, 0), ho(m * 96);
  bool any_addend = false;
  for (size_t q = 0; q < m; ++q) {
    const size_t j = sel[q];
    for (size_t t = offsets[j]; t < offsets[j + 1]; ++t) {
      const uint8_t* sc = term_scalars32 + 32 * t;
      bool unit = sc[0] <= 1;
      for (int b = 1; b < 32 && unit; ++b) unit = sc[b] == 0;
      const uint8_t* src = bases_affine96 + 96 * (size_t)(term_base[t] & 0x7fffffffu);
      if (!unit) {
        memcpy(&hb[96 * q], src, 96);
        if (term_base[t] >> 31) negate_affine96_y(&hb[96 * q]);
        memcpy(&hs[32 * q], sc, 32);
      } else if (sc[0] == 1) {
        memcpy(&ha[96 * q], src, 96);
        if (term_base[t] >> 31) negate_affine96_y(&ha[96 * q]);
        any_addend = true;
      }
    }
  }
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf db, ds, da, dout;
  HIPCHK(db.alloc(m * 96)); HIPCHK(ds.alloc(m * 32));
  HIPCHK(hipMemcpyAsync(db.p, hb.data(), m * 96, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipMemcpyAsync(ds.p, hs.data(), m * 32, hipMemcpyHostToDevice, ctx->stream));
  if (any_addend) { HIPCHK(da.alloc(m * 96)); HIPCHK(hipMemcpyAsync(da.p, ha.data(), m * 96, hipMemcpyHostToDevice, ctx->stream)); }
  if (ctx->batch_mul_row && m <= LINCOMB_ROW_MAX) {
    // one wave per result, one limb per lane; canonical XYZZ words come back (the host normalises all results of the batch together)
    HIPCHK(dout.alloc(m * sizeof(cg1::PointWords)));
    hipLaunchKernelGGL(cg1::k_batch_mul_row, dim3((unsigned)m), dim3(64), 0, ctx->stream, (const uint32_t*)db.p, (uint32_t)m, (const uint32_t*)ds.p, (uint32_t)m,
                       (const uint32_t*)da.p, (cg1::PointWords*)dout.p, (uint32_t)m);
    std::vector<cg1::PointWords> hw(m);
    HIPCHK(hipMemcpyAsync(hw.data(), dout.p, m * sizeof(cg1::PointWords), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    for (size_t q = 0; q < m; ++q) results[sel[q]] = cg1::jac_from_words(hw[q]);
    return CG1_OK;
  }
  HIPCHK(dout.alloc(m * 96));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int rc = cg1_batch_mul_add_device(ctx, db.p, m, ds.p, m, da.p, dout.p, m);
  if (rc != CG1_OK) return rc;
  HIPCHK(hipMemcpy(ho.data(), dout.p, m * 96, hipMemcpyDeviceToHost));
  for (size_t q = 0; q < m; ++q) {
    const uint8_t* rec = &ho[96 * q];
    bool zero = true;
    for (int k = 0; k < 96 && zero; ++k) zero = rec[k] == 0;
    if (zero) { results[sel[q]] = cg1h::jac_identity(); continue; }
    cg1h::fe x, y;
    if (!cg1h::fe_from_le48(rec, x) || !cg1h::fe_from_le48(rec + 48, y)) { snprintf(ctx->err, sizeof ctx->err, "k_batch_mul returned a non-canonical record"); return CG1_ERR_HIP; }
    results[sel[q]] = cg1h::jac_from_affine(x, y);
  }
  return CG1_OK;
}

extern "C" {
int cg1_lincomb_batch(cg1_ctx* ctx, const uint8_t* bases_affine96, size_t n_bases, const uint32_t* offsets, size_t n_out, const uint32_t* term_base,
                      const uint8_t* term_scalars32, int path, uint8_t* out_blobs144, uint8_t* out_affine96, uint8_t* out_comp48, int* path_used) {
  if (path_used) *path_used = 0;
  if (n_out == 0) return CG1_OK;
  if (!offsets || offsets[0] != 0 || path < 0 || path > 2) return CG1_ERR_ARG;
  const size_t T = offsets[n_out];
  if (T && (!bases_affine96 || !term_base || !term_scalars32)) return CG1_ERR_ARG;
  for (size_t j = 0; j < n_out; ++j) if (offsets[j] > offsets[j + 1]) return CG1_ERR_ARG;
  for (size_t t = 0; t < T; ++t) if ((term_base[t] & 0x7fffffffu) >= n_bases) { if (ctx) snprintf(ctx->err, sizeof ctx->err, "lincomb: base index out of range"); return CG1_ERR_ARG; }
  // ---- which outputs go where
  std::vector<uint32_t> gsel, psel;                          // output indices for the GPU / for the pool
  if (path == 1 || !ctx) {
    if (path == 2) return CG1_ERR_HIP;
    path = 1;
  } else if (path == 0) {
    auto gpu_est = [](size_t m, size_t max_terms) -> double {            // us
      if (m == 0) return 0.0;
      if (m <= cg1::SM_MAX_MSMS && max_terms <= cg1::SM_MAX_N) return 250.0 + 20.0 * (double)m;
      if (m <= LINCOMB_MAX_REGIME_B) return 1900.0 + 2.0 * (double)m;
      return 1e18;           // more independent MSMs than the regime-B chain has ever been run with: the pool (or k_batch_mul above) takes them
    };
    double ops_all = 0, ops_small = 0;
    size_t n_big = 0, big_max = 0, all_max = 0, n_shaped = 0;
    std::vector<uint8_t> big(n_out, 0);                    // 1: >= 4 weighted terms; 2: "s * B" or "A + s * B" (the callers' map / fold loops)
    for (size_t j = 0; j < n_out; ++j) {
      size_t heavy = 0, unit = 0;
      for (size_t t = offsets[j]; t < offsets[j + 1]; ++t) {
        const uint8_t* sc = term_scalars32 + 32 * t;
        bool small = sc[0] <= 1;
        for (int b = 1; b < 32 && small; ++b) small = sc[b] == 0;
        if (small) ++unit; else ++heavy;
      }
      const double ops = (heavy ? 255.0 : 0.0) + 52.0 * (double)heavy + (double)unit;
      ops_all += ops;
      all_max = std::max(all_max, (size_t)(offsets[j + 1] - offsets[j]));
      if (heavy >= 4) { big[j] = 1; ++n_big; big_max = std::max(big_max, (size_t)(offsets[j + 1] - offsets[j])); }
      else {
        ops_small += ops;
        if (heavy == 1 && offsets[j + 1] - offsets[j] <= 2) { big[j] = 2; ++n_shaped; }
      }
    }
    if (n_shaped >= 2048 || (ctx->batch_mul_row && n_shaped >= LINCOMB_ROW_MIN)) {
      // thousands of independent scalar multiplications (get_random_point over a long vector, a map / fold of 2^16 points): the batched
      // scalar-multiplication kernel (k_batch_mul: one lane per output, ~2.2 ms of dependent doublings whatever the count) takes them;
      // what is left of the batch is decided as below, without them
      std::vector<uint32_t> ssel;
      for (size_t j = 0; j < n_out; ++j) if (big[j] == 2) ssel.push_back((uint32_t)j);
      std::vector<cg1h::jac> all(n_out, cg1h::jac_identity());
      int rc = lincomb_shaped_device(ctx, bases_affine96, offsets, term_base, term_scalars32, ssel, all);
      if (rc != CG1_OK) return rc;
      if (ssel.size() < n_out) {
        // the rest as its own batch (recursion depth 1: no shaped outputs of this size are left in it)
        std::vector<uint32_t> rsel, roffs(1, 0), rtb;
        std::vector<uint8_t> rsc;
        for (size_t j = 0; j < n_out; ++j) if (big[j] != 2) {
          rsel.push_back((uint32_t)j);
          for (size_t t = offsets[j]; t < offsets[j + 1]; ++t) { rtb.push_back(term_base[t]); rsc.insert(rsc.end(), term_scalars32 + 32 * t, term_scalars32 + 32 * t + 32); }
          roffs.push_back((uint32_t)rtb.size());
        }
        std::vector<uint8_t> rblobs(rsel.size() * CG1_POINT_BYTES);
        rc = cg1_lincomb_batch(ctx, bases_affine96, n_bases, roffs.data(), rsel.size(), rtb.empty() ? nullptr : rtb.data(), rsc.empty() ? nullptr : rsc.data(), 0,
                               rblobs.data(), nullptr, nullptr, nullptr);
        if (rc != CG1_OK) return rc;
        for (size_t q = 0; q < rsel.size(); ++q) all[rsel[q]] = blob_in(rblobs.data() + CG1_POINT_BYTES * q);
      }
      if (path_used) *path_used = 2;
      cg1_lincomb_write_outputs(all.data(), n_out, out_blobs144, out_affine96, out_comp48);
      return CG1_OK;
    }
    const double pool_all = 0.25 * ops_all / (double)std::min<size_t>(n_out, 8);
    const double pool_small = n_out > n_big ? 0.25 * ops_small / (double)std::min<size_t>(n_out - n_big, 8) : 0.0;
    const double hybrid = std::max(gpu_est(n_big, big_max), pool_small) + (n_big && n_out > n_big ? 30.0 : 0.0);
    const double gpu_all = gpu_est(n_out, all_max);
    if (pool_all <= hybrid && pool_all <= gpu_all) path = 1;
    else if (gpu_all < hybrid || n_big == n_out) path = 2;
    else {
      path = 3;
      for (size_t j = 0; j < n_out; ++j) (big[j] == 1 ? gsel : psel).push_back((uint32_t)j);
    }
  }
  if (path_used) *path_used = path;
  if (path == 1) return cg1_lincomb_batch_pool(bases_affine96, n_bases, offsets, n_out, term_base, term_scalars32, out_blobs144, out_affine96, out_comp48, 0);
  if (path == 2 && n_out > LINCOMB_MAX_REGIME_B) {
    // the regime-B chain is run with at most LINCOMB_MAX_REGIME_B MSMs per call (what it has been measured with): halves
    const size_t h = n_out / 2;
    std::vector<uint32_t> o2(n_out - h + 1);
    for (size_t j = h; j <= n_out; ++j) o2[j - h] = offsets[j] - offsets[h];
    int rc = cg1_lincomb_batch(ctx, bases_affine96, n_bases, offsets, h, term_base, term_scalars32, 2, out_blobs144, out_affine96, out_comp48, nullptr);
    if (rc != CG1_OK) return rc;
    return cg1_lincomb_batch(ctx, bases_affine96, n_bases, o2.data(), n_out - h, term_base + offsets[h], term_scalars32 + 32 * (size_t)offsets[h], 2,
                             out_blobs144 ? out_blobs144 + CG1_POINT_BYTES * h : nullptr, out_affine96 ? out_affine96 + 96 * h : nullptr,
                             out_comp48 ? out_comp48 + 48 * h : nullptr, nullptr);
  }
  if (path == 2) { gsel.resize(n_out); for (size_t j = 0; j < n_out; ++j) gsel[j] = (uint32_t)j; }
  std::vector<cg1h::jac> res(n_out, cg1h::jac_identity());
  // ---- the GPU's share: its terms gathered (a negated base: y -> p - y on the standard-form record) into page-locked staging, one batched MSM
  const size_t G = gsel.size();
  std::vector<uint32_t> goffs(G + 1, 0);
  for (size_t q = 0; q < G; ++q) goffs[q + 1] = goffs[q] + (offsets[gsel[q] + 1] - offsets[gsel[q]]);
  const size_t TG = goffs[G];
  std::vector<cg1h::jac> gres;
  bool pending = false;
  if (TG) {
    HIPCHK(hipSetDevice(ctx->device));
    if (TG * 128 > ctx->cap_h_lin) {
      if (ctx->h_lin) (void)hipHostFree(ctx->h_lin);
      ctx->h_lin = nullptr; ctx->cap_h_lin = 0;
      const size_t want = TG * 128 + TG * 32 + 4096;
      HIPCHK(hipHostMalloc((void**)&ctx->h_lin, want, hipHostMallocDefault));
      ctx->cap_h_lin = want;
    }
    uint8_t* hp = ctx->h_lin;
    uint8_t* hs = ctx->h_lin + TG * 96;
    size_t o = 0;
    for (size_t q = 0; q < G; ++q) {
      for (size_t t = offsets[gsel[q]]; t < offsets[gsel[q] + 1]; ++t, ++o) {
        const uint8_t* src = bases_affine96 + 96 * (size_t)(term_base[t] & 0x7fffffffu);
        uint8_t* dst = hp + 96 * o;
        memcpy(dst, src, 96);
        if (term_base[t] >> 31) {
          uint64_t y[6], any = 0;
          memcpy(y, src + 48, 48);
          for (int i = 0; i < 6; ++i) any |= y[i];
          if (any) {                                         // (the identity record stays all-zero)
            unsigned __int128 br = 0;
            for (int i = 0; i < 6; ++i) { const unsigned __int128 d = (unsigned __int128)cg1::H_P[i] - y[i] - br; y[i] = (uint64_t)d; br = (d >> 64) & 1; }
            memcpy(dst + 48, y, 48);
          }
        }
        memcpy(hs + 32 * o, term_scalars32 + 32 * t, 32);
      }
    }
    { int src = ensure_stage(ctx, TG * 96, TG * 32); if (src) return src; }
    HIPCHK(hipMemcpyAsync(ctx->d_stage_pts, hp, TG * 96, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(ctx->d_stage_sc, hs, TG * 32, hipMemcpyHostToDevice, ctx->stream));
    int rc = cg1::msm_batched_device(ctx, ctx->d_stage_pts, ctx->d_stage_sc, goffs.data(), G, 0, gres, &pending);
    if (rc != CG1_OK) return rc;
  } else {
    gres.assign(G, cg1h::jac_identity());
  }
  // ---- the pool's share, while the launch runs
  int prc = 0;
  if (!psel.empty()) prc = cg1h::lincomb_pool_jac(bases_affine96, n_bases, offsets, term_base, term_scalars32, psel.data(), psel.size(), res.data(), 0);
  if (pending) { int rc = cg1::msm_batched_small_end(ctx, G, gres); if (rc != CG1_OK) return rc; }
  if (prc) return prc == 3 ? CG1_ERR_ENCODING : CG1_ERR_ARG;
  for (size_t q = 0; q < G; ++q) res[gsel[q]] = gres[q];
  cg1_lincomb_write_outputs(res.data(), n_out, out_blobs144, out_affine96, out_comp48);
  return CG1_OK;
}

int cg1_get_timings(const cg1_ctx* ctx, float* phase_ms, float* host_tail_ms, int* window_c) {
  if (!ctx) return CG1_ERR_ARG;
  if (phase_ms) for (int i = 0; i < CG1_NPHASE; ++i) phase_ms[i] = ctx->phase_ms[i];
  if (host_tail_ms) *host_tail_ms = ctx->host_tail_ms;
  if (window_c) *window_c = ctx->last_c;
  return CG1_OK;
}

int cg1_get_last_launches(const cg1_ctx* ctx) { return ctx ? ctx->last_acc_launches : -1; }

int cg1_get_last_counts(const cg1_ctx* ctx, uint32_t* entries, uint32_t* chunks) {
  if (!ctx) return CG1_ERR_ARG;
  if (entries) *entries = ctx->last_entries;
  if (chunks) *chunks = ctx->last_chunks;
  return CG1_OK;
}

// hipEvent stopwatch on the context's compute stream: everything enqueued between begin and end is timed on the device
int cg1_timer_begin(cg1_ctx* ctx) {
  if (!ctx) return CG1_ERR_HIP;
  HIPCHK(hipSetDevice(ctx->device));
  for (int i = 0; i < 2; ++i) if (!ctx->tm_ev[i]) HIPCHK(hipEventCreate(&ctx->tm_ev[i]));
  HIPCHK(hipEventRecord(ctx->tm_ev[0], ctx->stream));
  return CG1_OK;
}
int cg1_timer_end(cg1_ctx* ctx, float* ms) {
  if (!ctx || !ms || !ctx->tm_ev[0] || !ctx->tm_ev[1]) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipEventRecord(ctx->tm_ev[1], ctx->stream));
  HIPCHK(hipEventSynchronize(ctx->tm_ev[1]));
  HIPCHK(hipEventElapsedTime(ms, ctx->tm_ev[0], ctx->tm_ev[1]));
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
  if (ctx->quad && n <= (size_t)ctx->batch_mul_quad_max)       // latency-bound launches: one DPP quad per output
    hipLaunchKernelGGL(cg1::k_batch_mul_quad, dim3((unsigned)((n * 4 + 63) / 64)), dim3(64), 0, ctx->stream,
                       (const uint32_t*)d_bases, (uint32_t)nbase, (const uint32_t*)d_scalars, (uint32_t)nscalars,
                       (const uint32_t*)d_addend, (uint32_t*)d_out, (uint32_t)n);
  else
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
  {
    // one error contract whichever engine serves the call: every coordinate a canonical field element (< p); the curve equation is not checked
    auto canonical = [](const uint8_t* rec) {
      for (int c = 0; c < 2; ++c) {
        uint64_t w[6];
        memcpy(w, rec + 48 * c, 48);
        bool lt = false;
        for (int i = 5; i >= 0; --i) { if (w[i] != cg1::H_P[i]) { lt = w[i] < cg1::H_P[i]; break; } }
        if (!lt) return false;
      }
      return true;
    };
    for (size_t i = 0; i < nbase; ++i) if (!canonical(bases + 96 * i)) { snprintf(ctx->err, sizeof ctx->err, "base %zu: coordinate >= p", i); return CG1_ERR_ENCODING; }
    if (addend) for (size_t i = 0; i < n; ++i) if (!canonical(addend + 96 * i)) { snprintf(ctx->err, sizeof ctx->err, "addend %zu: coordinate >= p", i); return CG1_ERR_ENCODING; }
    // Which engine -- decided by the call alone, never by the machine ("batch_mul_host_max": -1 = this rule, 0 = never the host, N = the
    // host up to N outputs):  up to 96 outputs the host's pool (~77 us each over its threads against a ~0.6 ms launch);  up to 4 096 one
    // WAVE per output with one limb per lane (k_batch_mul_row: 255 doublings at a lone wave's ~1.5 us, ~0.55 ms whatever n is,
    // "batch_mul_row" = 0 switches it off);  beyond, one quad / one lane per output (k_batch_mul_quad / k_batch_mul: ~2.2 ms up to 8 192).
    const size_t host_max = ctx->batch_mul_host_max >= 0 ? (size_t)ctx->batch_mul_host_max : LINCOMB_ROW_MIN;
    ctx->last_batch_mul_on_host = 0;
    if (n <= host_max) {
      ctx->last_batch_mul_on_host = 1;
      return cg1_batch_mul_add_pool(bases, nbase, scalars, nscalars, addend, out, n, 0);
    }
    if (ctx->batch_mul_row && n <= LINCOMB_ROW_MAX) {
      HIPCHK(hipSetDevice(ctx->device));
      DevBuf db, ds, da, dout;
      HIPCHK(db.alloc(nbase * 96)); HIPCHK(ds.alloc(nscalars * 32)); HIPCHK(dout.alloc(n * sizeof(cg1::PointWords)));
      HIPCHK(hipMemcpyAsync(db.p, bases, nbase * 96, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(ds.p, scalars, nscalars * 32, hipMemcpyHostToDevice, ctx->stream));
      if (addend) { HIPCHK(da.alloc(n * 96)); HIPCHK(hipMemcpyAsync(da.p, addend, n * 96, hipMemcpyHostToDevice, ctx->stream)); }
      hipLaunchKernelGGL(cg1::k_batch_mul_row, dim3((unsigned)n), dim3(64), 0, ctx->stream, (const uint32_t*)db.p, (uint32_t)nbase, (const uint32_t*)ds.p,
                         (uint32_t)nscalars, (const uint32_t*)da.p, (cg1::PointWords*)dout.p, (uint32_t)n);
      std::vector<cg1::PointWords> hw(n);
      HIPCHK(hipMemcpyAsync(hw.data(), dout.p, n * sizeof(cg1::PointWords), hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(hipGetLastError());
      std::vector<cg1h::jac> res(n);
      for (size_t i = 0; i < n; ++i) res[i] = cg1::jac_from_words(hw[i]);
      cg1_lincomb_write_outputs(res.data(), n, nullptr, out, nullptr);      // ONE shared inversion on the host: affine96 records
      return CG1_OK;
    }
  }
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf db, ds, da, dout;
  HIPCHK(db.alloc(nbase * 96)); HIPCHK(ds.alloc(nscalars * 32)); HIPCHK(dout.alloc(n * 96));
  HIPCHK(hipMemcpy(db.p, bases, nbase * 96, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ds.p, scalars, nscalars * 32, hipMemcpyHostToDevice));
  if (addend) { HIPCHK(da.alloc(n * 96)); HIPCHK(hipMemcpy(da.p, addend, n * 96, hipMemcpyHostToDevice)); }
  int rc = cg1_batch_mul_add_device(ctx, db.p, nbase, ds.p, nscalars, da.p, dout.p, n);
  if (rc != CG1_OK) return rc;
  HIPCHK(hipMemcpy(out, dout.p, n * 96, hipMemcpyDeviceToHost));
  return CG1_OK;
}
}  // extern "C"
namespace {
void launch_decompress(cg1_ctx* ctx, const void* d_in48, void* d_out_affine96, void* d_status, size_t n, int check_subgroup) {
  const dim3 grid((unsigned)((n + 127) / 128)), block(128);
  if (check_subgroup)
    hipLaunchKernelGGL((cg1::k_batch_decompress<true, 2>), grid, block, 0, ctx->stream, (const uint8_t*)d_in48, (uint32_t*)d_out_affine96, (uint8_t*)d_status, (uint32_t)n);
  else if (ctx->decompress_waves == 3)
    hipLaunchKernelGGL((cg1::k_batch_decompress<false, 3>), grid, block, 0, ctx->stream, (const uint8_t*)d_in48, (uint32_t*)d_out_affine96, (uint8_t*)d_status, (uint32_t)n);
  else
    hipLaunchKernelGGL((cg1::k_batch_decompress<false, 2>), grid, block, 0, ctx->stream, (const uint8_t*)d_in48, (uint32_t*)d_out_affine96, (uint8_t*)d_status, (uint32_t)n);
}
}
extern "C" {
// n compressed48 (device) -> n affine96 + n status bytes (device); returns CG1_OK when the kernel ran
int cg1_batch_decompress_device(cg1_ctx* ctx, const void* d_in48, void* d_out_affine96, void* d_status, size_t n, int check_subgroup) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) return CG1_OK;
  if (n >= (1ull << 31)) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  launch_decompress(ctx, d_in48, d_out_affine96, d_status, n, check_subgroup);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  return CG1_OK;
}
// same launch as cg1_batch_decompress_device without waiting for it (pair with cg1_ctx_sync)
int cg1_batch_decompress_enqueue(cg1_ctx* ctx, const void* d_in48, void* d_out_affine96, void* d_status, size_t n, int check_subgroup) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) return CG1_OK;
  if (n >= (1ull << 31)) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  launch_decompress(ctx, d_in48, d_out_affine96, d_status, n, check_subgroup);
  HIPCHK(hipGetLastError());
  return CG1_OK;
}
// Subgroup flags of k selected points per proof (see k_subgroup_flags), on the context's SIDE stream: ordered after
// everything enqueued on the compute stream so far (the decompression that produced the points), running beside what is
// enqueued there next.  cg1_side_sync waits for it.
int cg1_subgroup_flags_enqueue(cg1_ctx* ctx, const void* d_affine96, size_t stride_points, size_t n_proofs,
                               const uint32_t* offsets, size_t k, void* d_flags) {
  if (!ctx) return CG1_ERR_HIP;
  if (n_proofs == 0 || k == 0) return CG1_OK;
  if (!offsets || k > 16 || n_proofs * k >= (1ull << 29)) return CG1_ERR_ARG;
  for (size_t j = 0; j < k; ++j) if (offsets[j] >= stride_points) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  if (!ctx->side_stream) {
    if (!ctx->cu_mask.empty()) HIPCHK(hipExtStreamCreateWithCUMask(&ctx->side_stream, (uint32_t)ctx->cu_mask.size(), ctx->cu_mask.data()));
    else HIPCHK(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ctx->side_ev, hipEventDisableTiming));
  }
  HIPCHK(hipEventRecord(ctx->side_ev, ctx->stream));
  HIPCHK(hipStreamWaitEvent(ctx->side_stream, ctx->side_ev, 0));
  cg1::SgOffsets so;
  for (size_t j = 0; j < 16; ++j) so.off[j] = j < k ? offsets[j] : 0u;
  so.k = (uint32_t)k;
  const size_t lanes = n_proofs * k * 4;           // one DPP quad per point
  hipLaunchKernelGGL(cg1::k_subgroup_flags, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, ctx->side_stream,
                     (const uint32_t*)d_affine96, (uint32_t)stride_points, (uint32_t)n_proofs, so, (uint8_t*)d_flags);
  HIPCHK(hipGetLastError());
  return CG1_OK;
}
int cg1_side_sync(cg1_ctx* ctx) {
  if (!ctx) return CG1_ERR_HIP;
  if (!ctx->side_stream) return CG1_OK;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->side_stream));
  return CG1_OK;
}
// n Merlin transcripts on the device, one per lane, all running the same operation list on their own data rows
// (k_merlin_batch).  init_state208: MerlinTranscript(label) as the host left it (cg1_merlin_init); ops: host array.
}  // extern "C"
namespace {
bool build_block_program(const std::vector<cg1merlin::COp>& ops, const std::vector<std::string>& labels, const uint8_t* init, const uint8_t* consts, bool generic,
                         std::vector<cg1merlin::RowDesc>& desc, uint32_t& n_nodes);      // (defined with the front-end's program below)
}
extern "C" {
int cg1_merlin_batch_device(cg1_ctx* ctx, const uint8_t* init_state208, const cg1_merlin_op* ops, size_t nops, const void* d_data,
                            size_t data_stride, void* d_out, size_t out_stride, void* d_states_out, size_t n) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) return CG1_OK;
  if (!init_state208 || (nops && !ops) || !d_out || n >= (1ull << 31)) return CG1_ERR_ARG;
  static_assert(sizeof(cg1_merlin_op) == sizeof(cg1merlin::Op), "op record layout");
  for (size_t k = 0; k < nops; ++k) {
    const cg1_merlin_op& o = ops[k];
    if (o.kind > 3 || o.label_len > 32) return CG1_ERR_ARG;
    if (o.kind == 0 && (!d_data || (size_t)o.data_off + o.len > data_stride)) return CG1_ERR_ARG;
    if (o.kind != 0 && (size_t)o.out_off + (o.kind == 2 ? 32 : o.len) > out_stride) return CG1_ERR_ARG;
  }
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf dst, dops;
  HIPCHK(dst.alloc(208)); HIPCHK(dops.alloc(nops * sizeof(cg1_merlin_op)));
  HIPCHK(hipMemcpyAsync(dst.p, init_state208, 208, hipMemcpyHostToDevice, ctx->stream));
  if (nops) HIPCHK(hipMemcpyAsync(dops.p, ops, nops * sizeof(cg1_merlin_op), hipMemcpyHostToDevice, ctx->stream));
  const unsigned nblk = (unsigned)((n + cg1merlin::LANES - 1) / cg1merlin::LANES);
  if (ctx->merlin_sync) {
    // the kernel's own records: 16 bytes per operation, the distinct labels in a table (it keeps them in LDS)
    std::vector<cg1merlin::COp> cops(nops);
    std::vector<uint32_t> table;
    std::vector<std::pair<std::vector<uint8_t>, uint32_t>> seen;
    bool fits = true;
    for (size_t k = 0; k < nops && fits; ++k) {
      const cg1_merlin_op& o = ops[k];
      std::vector<uint8_t> lb(o.label, o.label + o.label_len);
      uint32_t idx = (uint32_t)seen.size();
      for (const auto& e : seen) if (e.first == lb) { idx = e.second; break; }
      if (idx == seen.size()) {
        if (seen.size() >= (size_t)cg1merlin::MAX_LABELS) { fits = false; break; }
        seen.emplace_back(lb, idx);
        uint8_t padded[32] = {0};
        memcpy(padded, o.label, o.label_len);
        for (int j = 0; j < 8; ++j) { uint32_t v; memcpy(&v, padded + 4 * j, 4); table.push_back(v); }
      }
      cops[k] = cg1merlin::COp{(uint32_t)o.kind | (idx << 8) | ((uint32_t)o.label_len << 16), o.len, o.data_off, o.out_off};
    }
    if (fits && ctx->merlin_rows) {
      // the block program (kernels_merlin.h): whole rate blocks per pass; falls through to the byte machine when the program does not
      // fit the row format (a challenge longer than 164 bytes, more than four late pieces in a block, unaligned output offsets)
      std::vector<std::string> labels(seen.size());
      for (const auto& e : seen) labels[e.second] = std::string(e.first.begin(), e.first.end());
      std::vector<cg1merlin::RowDesc> desc;
      uint32_t nn = 0;
      const unsigned lanes_used = (unsigned)ctx->merlin_lanes;
      const unsigned nb = (unsigned)((n + lanes_used - 1) / lanes_used);
      const size_t need = build_block_program(cops, labels, init_state208, nullptr, true, desc, nn) ? (size_t)nb * lanes_used * nn * cg1merlin::ROW_WORDS * 4 : 0;
      if (need && need <= ((size_t)8 << 30)) {
        if (need > ctx->merlin_rows_cap) {
          if (ctx->d_merlin_rows) (void)hipFree(ctx->d_merlin_rows);
          ctx->d_merlin_rows = nullptr; ctx->merlin_rows_cap = 0;
          HIPCHK(hipMalloc(&ctx->d_merlin_rows, need));
          ctx->merlin_rows_cap = need;
        }
        DevBuf ddesc, dpass;
        HIPCHK(ddesc.alloc(desc.size() * sizeof(cg1merlin::RowDesc)));
        HIPCHK(dpass.alloc(4 * (size_t)nb));
        HIPCHK(hipMemcpyAsync(ddesc.p, desc.data(), desc.size() * sizeof(cg1merlin::RowDesc), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(cg1merlin::k_fill_rows, dim3((nn * cg1merlin::ROW_WORDS + 255u) / 256u, (unsigned)std::min<size_t>(n, 65535)), dim3(256), 0, ctx->stream, (const cg1merlin::RowDesc*)ddesc.p, nn,
                           (const uint8_t*)d_data, data_stride, 0u, (uint32_t)n, lanes_used, (uint32_t*)ctx->d_merlin_rows);
        hipLaunchKernelGGL(cg1merlin::k_merlin_batch_rows, dim3(nb), dim3(cg1merlin::LANES), 0, ctx->stream, (const uint8_t*)dst.p, (const uint32_t*)ctx->d_merlin_rows, nn,
                           (uint8_t*)d_out, out_stride, (uint8_t*)d_states_out, (uint32_t)n, lanes_used, (uint32_t*)dpass.p);
        HIPCHK(hipStreamSynchronize(ctx->stream));
        HIPCHK(hipGetLastError());
        std::vector<uint32_t> hp(nb);
        HIPCHK(hipMemcpy(hp.data(), dpass.p, 4 * (size_t)nb, hipMemcpyDeviceToHost));
        ctx->merlin_passes = *std::max_element(hp.begin(), hp.end());
        ctx->merlin_clk[0] = ctx->merlin_clk[1] = 0;
        ctx->merlin_last_kernel = 2;
        return CG1_OK;
      }
    }
    if (fits) {
      DevBuf dpass, dcops, dtab;
      const unsigned lanes_used = (unsigned)ctx->merlin_lanes;
      const unsigned nblk = (unsigned)((n + lanes_used - 1) / lanes_used);
      HIPCHK(dpass.alloc(16 * (size_t)nblk));
      HIPCHK(dcops.alloc(sizeof(cg1merlin::COp) * (nops ? nops : 1)));
      HIPCHK(dtab.alloc(4 * (table.size() ? table.size() : 8)));
      if (nops) HIPCHK(hipMemcpyAsync(dcops.p, cops.data(), sizeof(cg1merlin::COp) * nops, hipMemcpyHostToDevice, ctx->stream));
      if (!table.empty()) HIPCHK(hipMemcpyAsync(dtab.p, table.data(), 4 * table.size(), hipMemcpyHostToDevice, ctx->stream));
      hipLaunchKernelGGL(cg1merlin::k_merlin_batch_sync, dim3(nblk), dim3(cg1merlin::LANES), 0, ctx->stream,
                         (const uint8_t*)dst.p, (const cg1merlin::COp*)dcops.p, (uint32_t)nops, (const uint32_t*)dtab.p, (uint32_t)seen.size(),
                         (const uint8_t*)d_data, data_stride, (uint8_t*)d_out, out_stride, (uint8_t*)d_states_out, (uint32_t)n, (uint32_t*)dpass.p, lanes_used);
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(hipGetLastError());
      std::vector<uint32_t> hp(4 * (size_t)nblk);
      HIPCHK(hipMemcpy(hp.data(), dpass.p, 16 * (size_t)nblk, hipMemcpyDeviceToHost));
      ctx->merlin_passes = 0;
      for (unsigned b = 0; b < nblk; ++b)
        if (hp[4 * b] >= ctx->merlin_passes) { ctx->merlin_passes = hp[4 * b]; ctx->merlin_clk[0] = hp[4 * b + 1]; ctx->merlin_clk[1] = hp[4 * b + 2]; }
      if (getenv("CG1_MERLIN_TRACE"))
        fprintf(stderr, "k_merlin_batch_sync: %u passes; s_memtime ticks / 256 in advance %u, in Keccak %u (slowest wave)\n", ctx->merlin_passes, ctx->merlin_clk[0], ctx->merlin_clk[1]);
      ctx->merlin_last_kernel = 1;
      return CG1_OK;
    }
    // more than MAX_LABELS distinct labels: the round-2 kernel takes the program as it is
  }
  ctx->merlin_passes = 0;
  ctx->merlin_last_kernel = 0;
  hipLaunchKernelGGL(cg1merlin::k_merlin_batch, dim3(nblk), dim3(cg1merlin::LANES), 0, ctx->stream,
                     (const uint8_t*)dst.p, (const cg1merlin::Op*)dops.p, (uint32_t)nops, (const uint8_t*)d_data, data_stride,
                     (uint8_t*)d_out, out_stride, (uint8_t*)d_states_out, (uint32_t)n);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  return CG1_OK;
}
// Opening proofs, the batch verifier's front-end on the device (kernels_opening.h): the wire bytes of n proofs go up as they are, the
// five own points of each are gathered in MSM order and decompressed WITH the subgroup test (both equalities are asserted exactly by the
// reference, opening.py:73-74, on points it decodes unchecked: a random combination is sound only inside G1), the six-append transcript
// runs through the block program, and the scalars of the merged check are written behind one another: what the caller hands to
// cg1_msm_device is d_points96 / d_scalars32 with 5 n + 1 terms (the last one the generator with the summed scalar).  weights64 == NULL:
// the weights are derived on the device from seed32 (kernels_opening.h weights_from_seed; cg1_opening_weights_from_seed is the host's copy).  status[i] and
// point_status[5 i ..] come back exactly as cg1_opening_prepare + cg1_shuffle_apply_point_status leave them on the host path.
int cg1_opening_prepare_device(cg1_ctx* ctx, size_t n, const uint8_t* trackers96, const uint8_t* k_commitments48, const uint8_t* proofs128,
                               const uint8_t* weights64, const uint8_t* seed32, void* d_points96, void* d_scalars32, int32_t* status,
                               uint8_t* point_status, uint8_t* out_g_scalars32) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) return CG1_OK;
  if (!trackers96 || !k_commitments48 || !proofs128 || (!weights64 && !seed32) || !d_points96 || !d_scalars32 || !status || !point_status || n >= (1ull << 26))
    return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  // scratch: trackers 96 | k_commitments 48 | proofs 128 | weights 64 | wire 240 | rows 288 | challenges 32 | g scalars 32 | status 4 | point status 5 (+3)
  const size_t per = 96 + 48 + 128 + 64 + 240 + cg1open::ROW_BYTES + 32 + 32 + 4 + 8, need = per * n;
  if (need > ctx->cap_opening) {
    if (ctx->d_opening) (void)hipFree(ctx->d_opening);
    ctx->d_opening = nullptr; ctx->cap_opening = 0;
    HIPCHK(hipMalloc(&ctx->d_opening, need));
    ctx->cap_opening = need;
  }
  uint8_t* base = (uint8_t*)ctx->d_opening;
  uint8_t *d_trk = base, *d_kc = d_trk + 96 * n, *d_pf = d_kc + 48 * n, *d_w = d_pf + 128 * n, *d_wire = d_w + 64 * n, *d_rows = d_wire + 240 * n,
          *d_ch = d_rows + (size_t)cg1open::ROW_BYTES * n, *d_gs = d_ch + 32 * n, *d_st = d_gs + 32 * n, *d_ps = d_st + 4 * n;
  hipStream_t st = ctx->stream;
  HIPCHK(hipMemcpyAsync(d_trk, trackers96, 96 * n, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_kc, k_commitments48, 48 * n, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_pf, proofs128, 128 * n, hipMemcpyHostToDevice, st));
  if (weights64) HIPCHK(hipMemcpyAsync(d_w, weights64, 64 * n, hipMemcpyHostToDevice, st));
  cg1open::Seed32 seed{};
  if (!weights64) memcpy(seed.w, seed32, 32);
  uint8_t gblob[CG1_POINT_BYTES], g48[48], g96[96];
  cg1_generator(gblob);
  cg1_compress(g48, gblob);
  cg1_to_affine96(g96, gblob);
  cg1open::Enc48 genc;
  memcpy(genc.w, g48, 48);
  const uint32_t n32 = (uint32_t)n;
  hipLaunchKernelGGL(cg1open::k_opening_gather, dim3((unsigned)((6 * n + 255) / 256)), dim3(256), 0, st, (const uint32_t*)d_trk, (const uint32_t*)d_kc,
                     (const uint32_t*)d_pf, genc, n32, (uint32_t*)d_wire, (uint32_t*)d_rows);
  launch_decompress(ctx, d_wire, d_points96, d_ps, 5 * n, 1);
  HIPCHK(hipMemcpyAsync((uint8_t*)d_points96 + 96 * 5 * n, g96, 96, hipMemcpyHostToDevice, st));
  uint8_t init[CG1_MERLIN_STATE_BYTES];
  cg1_merlin_init(init, (const uint8_t*)"whisk_opening_proof", 19);                      // opening.py:60
  cg1_merlin_op ops[7];
  memset(ops, 0, sizeof ops);
  static const uint32_t off[6] = {0, 240, 48, 96, 144, 192};                              // k_G G k_r_G r_G A B (opening.py:61-66) inside a row
  for (int k = 0; k < 6; ++k) {
    ops[k].kind = 0; ops[k].label_len = 21; memcpy(ops[k].label, "tracker_opening_proof", 21);
    ops[k].len = 48; ops[k].data_off = off[k];
  }
  ops[6].kind = 2; ops[6].label_len = 31; memcpy(ops[6].label, "tracker_opening_proof_challenge", 31);
  { int rc = cg1_merlin_batch_device(ctx, init, ops, 7, d_rows, cg1open::ROW_BYTES, d_ch, 32, nullptr, n); if (rc) return rc; }
  hipLaunchKernelGGL(cg1open::k_opening_scalars, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (const uint8_t*)d_ch, (const uint8_t*)d_pf,
                     weights64 ? (const uint8_t*)d_w : (const uint8_t*)nullptr, seed, (const uint8_t*)d_ps, n32, (int32_t)CG1_SHUFFLE_BAD_SCALAR, (int32_t)CG1_SHUFFLE_BAD_WEIGHT, (int32_t)CG1_SHUFFLE_BAD_POINT,
                     (uint8_t*)d_scalars32, d_gs, (int32_t*)d_st);
  const unsigned sum_blocks = (unsigned)std::min<size_t>(256, (n + 1023) / 1024);          // the challenges are spent: their buffer takes the partial sums
  hipLaunchKernelGGL(cg1open::k_fr_sum, dim3(sum_blocks), dim3(256), 0, st, (const uint64_t*)d_gs, n32, (uint64_t*)d_ch);
  hipLaunchKernelGGL(cg1open::k_fr_sum, dim3(1), dim3(256), 0, st, (const uint64_t*)d_ch, sum_blocks, (uint64_t*)((uint8_t*)d_scalars32 + 32 * 5 * n));
  HIPCHK(hipMemcpyAsync(status, d_st, 4 * n, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(point_status, d_ps, 5 * n, hipMemcpyDeviceToHost, st));
  if (out_g_scalars32) HIPCHK(hipMemcpyAsync(out_g_scalars32, d_gs, 32 * n, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  return CG1_OK;
}
// The scalar rows of a batch of shuffle statements, built on the device from the host front-end's input blocks
// (cg1_shuffle_prepare_inputs), and the sum of the live proofs' CRS rows written behind the own-point scalars
// (d_out_scalars: n_proofs x (4 ell + 19 + 10 lg) scalars, then ell + 9).  Asynchronous on the compute stream.
}  // extern "C" (reopened below)

// ---------------------------------------------------------------- the shuffle verifier's front-end on the device (kernels_frontend.h)
struct cg1_shuffle_fe {
  int device = 0;
  cg1fe::Params pr{};
  uint32_t nops = 0, nlabels = 0;
  void *d_init = nullptr, *d_ops = nullptr, *d_labels = nullptr, *d_consts = nullptr, *d_tabG = nullptr, *d_tabH = nullptr;
  void *d_four = nullptr, *d_scratch = nullptr; size_t cap_n = 0;
  // the block program (kernels_frontend.h, second form): row descriptors per (node, word); the rows of a launch; passes per wave
  void *d_desc = nullptr, *d_rows = nullptr, *d_passes = nullptr; uint32_t n_nodes = 0; size_t cap_rows = 0, cap_blocks = 0, last_blocks = 0; uint32_t last_split[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  char err[200] = {0};
};

namespace {
// The verifier's transcript as an operation list: the mirror of prepare_one (csrc/shuffle_verify.cpp), which follows
// curdleproofs.py:176-180, same_perm.py:91-96, grand_prod.py:137-143, ipa.py:204-212 + :168-176, same_scalar.py:82-99,
// same_msm.py:194-206 + :158-173.  Challenges land in their slots of the row-input block (cg1rows::RowIn).
struct FeProgram {
  std::vector<cg1merlin::COp> ops;
  std::vector<uint32_t> table;
  std::vector<std::string> labels;
  uint32_t label(const char* s) {
    for (size_t i = 0; i < labels.size(); ++i) if (labels[i] == s) return (uint32_t)i;
    labels.emplace_back(s);
    uint8_t padded[32] = {0};
    memcpy(padded, s, strlen(s));
    for (int j = 0; j < 8; ++j) { uint32_t v; memcpy(&v, padded + 4 * j, 4); table.push_back(v); }
    return (uint32_t)labels.size() - 1;
  }
  void op(uint8_t kind, const char* lab, uint32_t len, uint32_t data_off, uint32_t out_off) {
    const uint32_t li = lab ? label(lab) : 0u, ll = lab ? (uint32_t)strlen(lab) : 0u;
    ops.push_back(cg1merlin::COp{(uint32_t)kind | (li << 8) | (ll << 16), len, data_off, out_off});
  }
  void point(const char* lab, size_t idx) { op(cg1merlin::OP_APPEND_POINT, lab, 48, (uint32_t)(idx * 48), 0); }
  void out(const char* lab, size_t slot, uint32_t len) { op(cg1merlin::OP_APPEND_OUT, lab, len, 0, (uint32_t)(slot * 32)); }
  void cst(const char* lab, uint32_t off) { op(cg1merlin::OP_APPEND_CONST, lab, 48, off, 0); }
  void challenge(const char* lab, size_t slot) { op(cg1merlin::OP_CHALLENGE_SCALAR, lab, 32, 0, (uint32_t)(slot * 32)); }
};

void fe_build_program(size_t ell, size_t lg, FeProgram& P) {
  const cg1rows::RowIn R{ell, lg};
  const size_t K = R.count(), base = 4 * ell;
  // own-point indices (csrc/shuffle_verify.cpp Layout)
  const size_t M = base, A = base + 1, T1 = base + 2, T2 = base + 3, U1 = base + 4, U2 = base + 5, Rp = base + 6, Sp = base + 7, B = base + 8, C = base + 9,
               Bc = base + 10, Bd = base + 11, LC = base + 12, RC = LC + lg, LD = LC + 2 * lg, RD = LC + 3 * lg, cmA1 = base + 12 + 4 * lg,
               Ba = cmA1 + 4, Bt = cmA1 + 5, Bu = cmA1 + 6, LA = cmA1 + 7, LT = LA + lg, LU = LA + 2 * lg, RA = LA + 3 * lg, RT = LA + 4 * lg, RU = LA + 5 * lg;
  for (size_t i = 0; i < 4 * ell; ++i) P.point("curdleproofs_step1", i);
  P.point("curdleproofs_step1", M);
  for (size_t i = 0; i < ell; ++i) P.challenge("curdleproofs_vec_a", R.a() + i);
  P.point("same_perm_step1", A); P.point("same_perm_step1", M);
  for (size_t i = 0; i < ell; ++i) P.out("same_perm_step1", R.a() + i, 32);
  P.challenge("same_perm_alpha", R.head() + 0); P.challenge("same_perm_beta", R.head() + 1);
  P.op(cg1fe::X_GPROD, nullptr, 0, 0, 0);
  P.point("gprod_step1", B); P.out("gprod_step1", K + 0, 32);
  P.challenge("gprod_alpha", R.head() + 2);
  P.point("gprod_step2", C); P.out("gprod_step2", K + 1, 32);
  P.challenge("gprod_beta", R.head() + 3);
  P.op(cg1fe::X_DA, nullptr, 0, 0, 0);
  P.point("ipa_step1", C); P.out("ipa_step1", K + 2, 48); P.out("ipa_step1", R.inner_prod(), 32); P.point("ipa_step1", Bc); P.point("ipa_step1", Bd);
  P.challenge("ipa_alpha", R.head() + 4); P.challenge("ipa_beta", R.head() + 5);
  for (size_t j = 0; j < lg; ++j) {
    P.point("ipa_loop", LC + j); P.point("ipa_loop", LD + j); P.point("ipa_loop", RC + j); P.point("ipa_loop", RD + j);
    P.challenge("ipa_gamma", R.gam() + j);
  }
  {
    const size_t order[10] = {Rp, Sp, T1, T2, U1, U2, cmA1, cmA1 + 1, cmA1 + 2, cmA1 + 3};
    for (size_t k = 0; k < 10; ++k) P.point("sameexp_points", order[k]);
  }
  P.challenge("same_scalar_alpha", R.head() + 6);
  P.out("same_msm_step1", K + 4, 48); P.point("same_msm_step1", T2); P.point("same_msm_step1", U2);
  for (size_t i = 0; i < ell; ++i) P.point("same_msm_step1", 2 * ell + i);
  P.cst("same_msm_step1", 0); P.cst("same_msm_step1", 0); P.cst("same_msm_step1", 48); P.cst("same_msm_step1", 0);       // Z Z H Z
  for (size_t i = 0; i < ell; ++i) P.point("same_msm_step1", 3 * ell + i);
  P.cst("same_msm_step1", 0); P.cst("same_msm_step1", 0); P.cst("same_msm_step1", 0); P.cst("same_msm_step1", 48);       // Z Z Z H
  P.point("same_msm_step1", Ba); P.point("same_msm_step1", Bt); P.point("same_msm_step1", Bu);
  P.challenge("same_msm_alpha", R.head() + 7);
  for (size_t j = 0; j < lg; ++j) {
    P.point("same_msm_loop", LA + j); P.point("same_msm_loop", LT + j); P.point("same_msm_loop", LU + j);
    P.point("same_msm_loop", RA + j); P.point("same_msm_loop", RT + j); P.point("same_msm_loop", RU + j);
    P.challenge("same_msm_gamma", R.gm() + j);
  }
  P.op(cg1fe::X_FINAL, nullptr, 0, 0, 0);
}

// The operation list cut into the nodes of kernels_frontend.h's block program: a symbolic run of STROBE (strobe.py:55-107) and of
// Merlin's framing (merlin_transcript.py:11-24, curdleproofs_transcript.py:15-25) that keeps, per byte of the sponge's rate, the
// constant XOR-ed into it and / or the place the byte comes from.  false = the program does not fit the row format (more than
// MAX_PIECES late pieces in a node, an offset too large): the caller keeps the byte-machine kernel.
struct FeNodes {
  struct Byte { uint8_t kind = 0; uint32_t src = 0; };              // 0 none, 1 byte of the lane's data row (src = its offset), 2 the challenge just drawn, 3 the out row
  struct Node { uint8_t T[168]; Byte D[168]; uint32_t type = cg1fe::N_PLAIN, bar = 0, da = 1, dr = 0, out_off = 0, len = 0; Node() { memset(T, 0, sizeof T); } };
  std::vector<Node> nodes;
  Node cur;
  uint32_t pos = 0, pos_begin = 0, cur_flags = 0;
  bool ok = true;
  int last_closed = -1;

  void run_f() {
    cur.T[pos] ^= (uint8_t)pos_begin; cur.T[pos + 1] ^= 0x04; cur.T[cg1merlin::STROBE_R + 1] ^= 0x80;
    nodes.push_back(cur);
    last_closed = (int)nodes.size() - 1;
    cur = Node();
    pos = 0; pos_begin = 0;
  }
  void put(uint8_t v) { cur.T[pos] ^= v; if (++pos == (uint32_t)cg1merlin::STROBE_R) run_f(); }
  void put_src(uint8_t kind, uint32_t src) { cur.D[pos].kind = kind; cur.D[pos].src = src; if (++pos == (uint32_t)cg1merlin::STROBE_R) run_f(); }
  void begin_op(uint8_t flags) {
    const uint32_t old = pos_begin;
    pos_begin = pos + 1;
    cur_flags = flags;
    put((uint8_t)old); put(flags);
    if ((flags & (cg1merlin::FLAG_C | cg1merlin::FLAG_K)) && pos != 0) run_f();
  }
  void frame(const std::string& label, uint32_t len) {
    begin_op(cg1merlin::FLAG_M | cg1merlin::FLAG_A);
    for (char c : label) put((uint8_t)c);
    for (int j = 0; j < 4; ++j) put((uint8_t)(len >> (8 * j)));
  }
  void barrier(uint32_t kind) { if (cur.bar) ok = false; cur.bar = kind; }
};

bool build_block_program(const std::vector<cg1merlin::COp>& ops, const std::vector<std::string>& labels, const uint8_t* init, const uint8_t* consts, bool generic,
                         std::vector<cg1merlin::RowDesc>& desc, uint32_t& n_nodes) {
  using namespace cg1merlin;
  FeNodes S;
  S.pos = init[200]; S.pos_begin = init[201]; S.cur_flags = init[202];
  if (S.pos >= (uint32_t)STROBE_R) return false;
  const uint32_t max_pieces = generic ? 4u : MAX_PIECES;            // (generic rows keep word 47 for the out-row offset / the final position)
  for (const COp& op : ops) {
    const uint32_t kind = op.kind_label & 0xffu, lab = (op.kind_label >> 8) & 0xffu, llen = op.kind_label >> 16;
    if (kind >= OP_BARRIER) {
      if (generic) return false;
      S.barrier(kind == cg1fe::X_GPROD ? 1u : (kind == cg1fe::X_DA ? 2u : 3u));
      continue;
    }
    if (lab >= labels.size()) return false;
    const std::string label = labels[lab].substr(0, llen);
    if (kind == OP_CHALLENGE_SCALAR || kind == OP_CHALLENGE) {
      const bool scalar = kind == OP_CHALLENGE_SCALAR;
      const uint32_t len = scalar ? 32u : op.len;
      if (len > 164u || (op.out_off & 3u) || (!scalar && !generic)) return false;
      S.frame(label, len);
      S.begin_op(FLAG_I | FLAG_A | FLAG_C);                              // the permutation the C flag forces closes the node the draw follows
      if (S.pos != 0 || S.last_closed < 0) return false;
      FeNodes::Node& sq = S.nodes[S.last_closed];
      if (sq.type != N_PLAIN) return false;
      sq.out_off = op.out_off; sq.len = len;
      if (!scalar) { sq.type = N_SQUEEZE_RAW; sq.da = 1; sq.dr = 0; S.pos = len; S.pos_begin = 0; continue; }
      sq.type = N_SQUEEZE; sq.da = 2; sq.dr = 1;
      // the redo node: the same frame and PRF header from (pos, pos_begin) = (32, 0), where every draw leaves the sponge
      S.pos = 32; S.pos_begin = 0;
      const size_t before = S.nodes.size();
      S.frame(label, 32);
      S.begin_op(FLAG_I | FLAG_A | FLAG_C);
      if (S.nodes.size() != before + 1 || S.pos != 0) return false;
      FeNodes::Node& rd = S.nodes.back();
      rd.type = N_SQUEEZE; rd.da = 1; rd.dr = 0; rd.out_off = op.out_off; rd.len = 32;
      // accepted: append_message(label, the 32 bytes), again from (32, 0)
      S.pos = 32; S.pos_begin = 0;
      S.frame(label, 32);
      S.begin_op(FLAG_A);
      for (uint32_t k = 0; k < 32; ++k) S.put_src(2, k);
      continue;
    }
    if (kind != OP_APPEND && kind != OP_APPEND_POINT && kind != OP_APPEND_CONST && kind != OP_APPEND_OUT) return false;
    if (kind == OP_APPEND_CONST && !consts) return false;
    S.frame(label, op.len);
    S.begin_op(FLAG_A);
    for (uint32_t k = 0; k < op.len; ++k) {
      if (kind == OP_APPEND_CONST) S.put(consts[op.data_off + k]);
      else if (kind == OP_APPEND_OUT) S.put_src(3, op.out_off + k);
      else S.put_src(1, op.data_off + k);
    }
  }
  S.cur.type = N_END;                                                     // what is left in the open node is never permuted
  S.cur.out_off = S.pos | (S.pos_begin << 8) | (S.cur_flags << 16);
  S.nodes.push_back(S.cur);
  if (!S.ok) return false;
  n_nodes = (uint32_t)S.nodes.size();
  desc.assign((size_t)n_nodes * ROW_WORDS, RowDesc{0, 0});
  for (uint32_t nd = 0; nd < n_nodes; ++nd) {
    const FeNodes::Node& N = S.nodes[nd];
    RowDesc* row = desc.data() + (size_t)nd * ROW_WORDS;
    for (uint32_t j = 0; j < 42; ++j) {
      uint32_t tw = 0;
      for (int b = 0; b < 4; ++b) tw |= (uint32_t)N.T[4 * j + b] << (8 * b);
      row[j].tword = tw;
      // data-row bytes of this word: one run of consecutive source bytes (a message is framed by >= 8 constant bytes)
      int lo = -1, cnt = 0;
      for (int b = 0; b < 4; ++b) if (N.D[4 * j + b].kind == 1) { if (lo < 0) lo = b; ++cnt; }
      if (cnt) {
        const uint32_t s0 = N.D[4 * j + lo].src;
        for (int b = 0; b < cnt; ++b) if (N.D[4 * j + lo + b].kind != 1 || N.D[4 * j + lo + b].src != s0 + b) return false;
        if (s0 >= (1u << 27) || (!generic && s0 % 48u + cnt > 48u)) return false;
        row[j].src = 1u | ((uint32_t)lo << 1) | ((uint32_t)(cnt - 1) << 3) | (s0 << 5);
      }
    }
    if (generic) {
      row[42].tword = N.type | (N.da << 4) | (N.dr << 6) | ((N.type == N_SQUEEZE_RAW ? N.len : 0u) << 8);
      row[47].tword = N.out_off;
    } else {
      if (N.type != N_END && ((N.out_off & 31u) || (N.out_off >> 5) >= (1u << 16))) return false;
      row[42].tword = N.type == N_END ? (N_END | (N.bar << 2)) : (N.type | (N.bar << 2) | (N.da << 4) | (N.dr << 6) | ((N.out_off >> 5) << 8));
    }
    if (N.type == N_END && !generic) continue;
    // late pieces: runs of bytes of kind 2 / 3 with consecutive sources
    uint32_t np = 0;
    for (uint32_t p = 0; p < (uint32_t)STROBE_R;) {
      const uint8_t k = N.D[p].kind;
      if (k < 2) { ++p; continue; }
      uint32_t len = 1;
      while (p + len < (uint32_t)STROBE_R && len < 48u && N.D[p + len].kind == k && N.D[p + len].src == N.D[p].src + len) ++len;
      if (np == max_pieces || N.D[p].src >= (1u << 17)) return false;
      row[43 + np].tword = len | (p << 6) | ((k == 3 ? 1u : 0u) << 14) | (N.D[p].src << 15);
      ++np;
      p += len;
    }
  }
  return true;
}
bool fe_build_nodes(const FeProgram& P, const uint8_t* init, const uint8_t* consts, std::vector<cg1fe::RowDesc>& desc, uint32_t& n_nodes) {
  return build_block_program(P.ops, P.labels, init, consts, false, desc, n_nodes);
}
}  // namespace

extern "C" {

void cg1_shuffle_fe_destroy(cg1_shuffle_fe* fe) {
  if (!fe) return;
  (void)hipSetDevice(fe->device);
  for (void* p : {fe->d_init, fe->d_ops, fe->d_labels, fe->d_consts, fe->d_tabG, fe->d_tabH, fe->d_four, fe->d_scratch, fe->d_desc, fe->d_rows, fe->d_passes})
    if (p) (void)hipFree(p);
  delete fe;
}

// crs_affine96 / crs48: the ell + 9 CRS points (crs.py:92-101 order) decoded and as they stand on the wire
cg1_shuffle_fe* cg1_shuffle_fe_create(cg1_ctx* ctx, size_t ell, size_t lg, const uint8_t* crs_affine96, const uint8_t* crs48) {
  if (!ctx || !crs_affine96 || !crs48 || ell == 0 || lg == 0 || lg > 20 || ((ell + 4) != ((size_t)1 << lg))) return nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
  cg1_shuffle_fe* fe = new cg1_shuffle_fe();
  fe->device = ctx->device;
  const cg1rows::RowIn R{ell, lg};
  cg1fe::Params& pr = fe->pr;
  pr.ell = (uint32_t)ell; pr.lg = (uint32_t)lg; pr.L = (uint32_t)(4 * ell + 19 + 10 * lg); pr.K = (uint32_t)R.count();
  pr.out_stride = (pr.K + 6u) * 32u;
  pr.idx_A = (uint32_t)(4 * ell + 1); pr.idx_T1 = (uint32_t)(4 * ell + 2); pr.idx_U1 = (uint32_t)(4 * ell + 4); pr.idx_B = (uint32_t)(4 * ell + 8);
  pr.idx_T0 = (uint32_t)(2 * ell);
  FeProgram P;
  fe_build_program(ell, lg, P);
  if (P.labels.size() > (size_t)cg1merlin::MAX_LABELS) { delete fe; return nullptr; }
  fe->nops = (uint32_t)P.ops.size(); fe->nlabels = (uint32_t)P.labels.size();
  uint8_t init[CG1_MERLIN_STATE_BYTES];
  cg1_merlin_init(init, (const uint8_t*)"curdleproofs", 12);                      // CurdleproofsTranscript(b"curdleproofs"), curdleproofs.py:172
  uint8_t consts[96];
  memset(consts, 0, sizeof consts);
  consts[0] = 0xC0;                                                               // Z1 as the wheel serialises it
  memcpy(consts + 48, crs48 + (ell + 4) * 48, 48);                                // crs.H
  bool ok = hipMalloc(&fe->d_init, sizeof init) == hipSuccess && hipMalloc(&fe->d_ops, P.ops.size() * sizeof(cg1merlin::COp)) == hipSuccess &&
            hipMalloc(&fe->d_labels, P.table.size() * 4) == hipSuccess && hipMalloc(&fe->d_consts, sizeof consts) == hipSuccess &&
            hipMalloc(&fe->d_tabG, 8192 * sizeof(cg1::PreparedPoint)) == hipSuccess && hipMalloc(&fe->d_tabH, 8192 * sizeof(cg1::PreparedPoint)) == hipSuccess;
  ok = ok && hipMemcpy(fe->d_init, init, sizeof init, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(fe->d_ops, P.ops.data(), P.ops.size() * sizeof(cg1merlin::COp), hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(fe->d_labels, P.table.data(), P.table.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
       hipMemcpy(fe->d_consts, consts, sizeof consts, hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    std::vector<cg1fe::RowDesc> desc;
    uint32_t nn = 0;
    if (fe_build_nodes(P, init, consts, desc, nn)) {
      ok = hipMalloc(&fe->d_desc, desc.size() * sizeof(cg1fe::RowDesc)) == hipSuccess &&
           hipMemcpy(fe->d_desc, desc.data(), desc.size() * sizeof(cg1fe::RowDesc), hipMemcpyHostToDevice) == hipSuccess;
      fe->n_nodes = nn;
    }
  }
  // fixed-base tables: entry [w][b] = b * 2^(8 w) * base for the two bases of D (grand_prod.py:157), as 128-byte Montgomery records
  if (ok) {
    std::vector<uint8_t> sc(8192 * 32, 0);
    for (int w = 0; w < 32; ++w) for (int b = 0; b < 256; ++b) sc[((size_t)w * 256 + b) * 32 + w] = (uint8_t)b;
    DevBuf dsc, dbase, dout, dflags;
    ok = dsc.alloc(sc.size()) == hipSuccess && dbase.alloc(96) == hipSuccess && dout.alloc(8192 * 96) == hipSuccess && dflags.alloc(8192 + 16) == hipSuccess &&
         hipMemcpy(dsc.p, sc.data(), sc.size(), hipMemcpyHostToDevice) == hipSuccess;
    for (int which = 0; which < 2 && ok; ++which) {
      const uint8_t* src = crs_affine96 + (ell + 4 + 3 + which) * 96;              // G_sum, H_sum
      ok = hipMemcpy(dbase.p, src, 96, hipMemcpyHostToDevice) == hipSuccess &&
           cg1_batch_mul_device(ctx, dbase.p, 1, dsc.p, dout.p, 8192) == CG1_OK;
      if (ok) {
        hipLaunchKernelGGL(cg1::k_prepare_points, dim3(32), dim3(256), 0, ctx->stream, (const uint32_t*)dout.p,
                           (cg1::PreparedPoint*)(which ? fe->d_tabH : fe->d_tabG), (uint8_t*)dflags.p, 8192u);
        ok = hipStreamSynchronize(ctx->stream) == hipSuccess && hipGetLastError() == hipSuccess;
      }
    }
  }
  if (!ok) { cg1_shuffle_fe_destroy(fe); return nullptr; }
  return fe;
}

size_t cg1_shuffle_fe_aux_bytes(void) { return 19 * 32; }

// Host only (no GPU needed; test support): ONE transcript of cg1_merlin_batch_device's interface run through the block program on the
// CPU -- build_block_program's tables walked as k_fill_rows + k_merlin_batch_rows walk them (rows, late pieces, first-draw / redo /
// raw squeeze nodes, the open block at the end).  Returns CG1_ERR_ARG when the operation list does not fit the row format (the device
// entry point then serves the call with the byte-level machine).  state_out208 may be NULL.
int cg1_merlin_block_program_emulate(const uint8_t* init_state208, const cg1_merlin_op* ops, size_t nops, const uint8_t* data_row, size_t data_bytes,
                                     uint8_t* out_row, size_t out_bytes, uint8_t* state_out208, uint32_t* passes) {
  if (!init_state208 || (nops && !ops) || !out_row) return CG1_ERR_ARG;
  std::vector<cg1merlin::COp> cops(nops);
  std::vector<std::string> labels;
  for (size_t k = 0; k < nops; ++k) {
    const cg1_merlin_op& o = ops[k];
    if (o.kind > 3 || o.label_len > 32) return CG1_ERR_ARG;
    if (o.kind == 0 && (size_t)o.data_off + o.len > data_bytes) return CG1_ERR_ARG;
    if (o.kind != 0 && (size_t)o.out_off + (o.kind == 2 ? 32 : o.len) > out_bytes) return CG1_ERR_ARG;
    const std::string lb((const char*)o.label, o.label_len);
    size_t idx = 0;
    while (idx < labels.size() && labels[idx] != lb) ++idx;
    if (idx == labels.size()) { if (labels.size() >= (size_t)cg1merlin::MAX_LABELS) return CG1_ERR_ARG; labels.push_back(lb); }
    cops[k] = cg1merlin::COp{(uint32_t)o.kind | ((uint32_t)idx << 8) | ((uint32_t)o.label_len << 16), o.len, o.data_off, o.out_off};
  }
  std::vector<cg1merlin::RowDesc> desc;
  uint32_t nn = 0;
  if (!build_block_program(cops, labels, init_state208, nullptr, true, desc, nn)) return CG1_ERR_ARG;
  uint8_t sponge[200], drawn[36] = {0};
  memcpy(sponge, init_state208, 200);
  uint32_t nd = 0, np = 0;
  for (;;) {
    const cg1merlin::RowDesc* row = desc.data() + (size_t)nd * cg1merlin::ROW_WORDS;
    const uint32_t info = row[42].tword, type = info & 3u, aux = row[47].tword;
    for (uint32_t q = 0; q < 4; ++q) {
      const uint32_t pc = row[43 + q].tword, len = pc & 63u;
      if (!len) continue;
      const uint32_t dst = (pc >> 6) & 255u, from_row = (pc >> 14) & 1u, so = pc >> 15;
      for (uint32_t i = 0; i < len; ++i) sponge[dst + i] ^= from_row ? out_row[so + i] : drawn[so + i];
    }
    for (uint32_t j = 0; j < 42; ++j) {
      uint32_t v = row[j].tword;
      if (row[j].src) {
        const uint32_t lo = (row[j].src >> 1) & 3u, cnt = ((row[j].src >> 3) & 3u) + 1u, off = row[j].src >> 5;
        for (uint32_t b = 0; b < cnt; ++b) v ^= (uint32_t)data_row[off + b] << (8 * (lo + b));
      }
      for (int b = 0; b < 4; ++b) sponge[4 * j + b] ^= (uint8_t)(v >> (8 * b));
    }
    if (type == cg1merlin::N_END) {
      if (state_out208) { memcpy(state_out208, sponge, 200); state_out208[200] = (uint8_t)aux; state_out208[201] = (uint8_t)(aux >> 8); state_out208[202] = (uint8_t)(aux >> 16); memset(state_out208 + 203, 0, 5); }
      break;
    }
    cg1_keccak_f1600(sponge);
    ++np;
    bool accept = true;
    if (type == cg1merlin::N_SQUEEZE) {
      uint8_t dv[32];
      memcpy(dv, sponge, 32);
      memset(sponge, 0, 32);
      cg1fr::fr tmp;
      bool nonzero = false;
      for (int i = 0; i < 32; ++i) nonzero |= dv[i] != 0;
      accept = nonzero && cg1fr::fr_from_le32(dv, tmp);
      if (accept) { memcpy(out_row + aux, dv, 32); memcpy(drawn, dv, 32); }
    } else if (type == cg1merlin::N_SQUEEZE_RAW) {
      const uint32_t len = (info >> 8) & 0xffu;
      memcpy(out_row + aux, sponge, len);
      memset(sponge, 0, len);
    }
    nd += accept ? (info >> 4) & 3u : (info >> 6) & 3u;
  }
  if (passes) *passes = np;
  return CG1_OK;
}

// Host only (no GPU needed; test support): walk the block program of one proof on the CPU exactly as k_shuffle_front_end_rows does --
// rows as k_fill_rows builds them, late pieces, both kinds of squeeze node -- up to the first barrier step (the grand product),
// and return the out row (the challenges drawn so far sit in their slots: vec_a, alpha, beta of same_perm).  wire = the proof's L own
// points as cg1_shuffle_gather_points packs them; out_row: (K + 6) * 32 bytes, zero where nothing was drawn.  *passes = permutations.
int cg1_shuffle_fe_emulate_to_first_barrier(size_t ell, size_t lg, const uint8_t* crs_h48, const uint8_t* wire, uint8_t* out_row, size_t out_row_bytes,
                                            uint32_t* passes) {
  if (!wire || !out_row || !crs_h48 || ell == 0 || lg == 0 || lg > 20 || ((ell + 4) != ((size_t)1 << lg))) return CG1_ERR_ARG;
  const cg1rows::RowIn R{ell, lg};
  if (out_row_bytes < (R.count() + 6) * 32) return CG1_ERR_ARG;
  FeProgram P;
  fe_build_program(ell, lg, P);
  uint8_t init[CG1_MERLIN_STATE_BYTES], consts[96];
  cg1_merlin_init(init, (const uint8_t*)"curdleproofs", 12);
  memset(consts, 0, sizeof consts);
  consts[0] = 0xC0;
  memcpy(consts + 48, crs_h48, 48);
  std::vector<cg1fe::RowDesc> desc;
  uint32_t nn = 0;
  if (P.labels.size() > (size_t)cg1merlin::MAX_LABELS || !fe_build_nodes(P, init, consts, desc, nn)) return CG1_ERR_ARG;
  memset(out_row, 0, out_row_bytes);
  uint8_t sponge[200], drawn[32] = {0};
  memcpy(sponge, init, 200);
  uint32_t nd = 0, np = 0;
  for (;;) {
    const cg1fe::RowDesc* row = desc.data() + (size_t)nd * cg1fe::ROW_WORDS;
    const uint32_t info = row[42].tword, type = info & 3u;
    if (((info >> 2) & 3u) != 0u || type == cg1fe::N_END) break;
    for (uint32_t q = 0; q < cg1fe::MAX_PIECES; ++q) {
      const uint32_t pc = row[43 + q].tword, len = pc & 63u;
      if (!len) break;
      const uint32_t dst = (pc >> 6) & 255u, from_row = (pc >> 14) & 1u, so = pc >> 15;
      for (uint32_t i = 0; i < len; ++i) sponge[dst + i] ^= from_row ? out_row[so + i] : drawn[so + i];
    }
    for (uint32_t j = 0; j < 42; ++j) {                      // the row as k_fill_rows writes it
      uint32_t v = row[j].tword;
      if (row[j].src) {
        const uint32_t lo = (row[j].src >> 1) & 3u, cnt = ((row[j].src >> 3) & 3u) + 1u, off = row[j].src >> 5, k0 = off % 48u;
        const uint8_t* pt = wire + (size_t)(off - k0);
        const bool inf = (pt[0] & 0xC0u) == 0xC0u;
        for (uint32_t b = 0; b < cnt; ++b) v ^= (uint32_t)(inf ? (k0 + b == 0 ? 0xC0u : 0u) : pt[k0 + b]) << (8 * (lo + b));
      }
      for (int b = 0; b < 4; ++b) sponge[4 * j + b] ^= (uint8_t)(v >> (8 * b));
    }
    cg1_keccak_f1600(sponge);
    ++np;
    bool accept = true;
    if (type == cg1fe::N_SQUEEZE) {
      uint8_t dv[32];
      memcpy(dv, sponge, 32);
      memset(sponge, 0, 32);
      cg1fr::fr tmp;
      bool nonzero = false;
      for (int i = 0; i < 32; ++i) nonzero |= dv[i] != 0;
      accept = nonzero && cg1fr::fr_from_le32(dv, tmp);
      if (accept) { memcpy(out_row + 32 * ((info >> 8) & 0xffffu), dv, 32); memcpy(drawn, dv, 32); }
    }
    nd += accept ? (info >> 4) & 3u : (info >> 6) & 3u;
  }
  if (passes) *passes = np;
  return CG1_OK;
}

// Host only (no GPU needed): the shape of the block program for a given ell -- operations of the verifier's transcript, nodes they are
// cut into (0: does not fit the row format), squeeze nodes (two per challenge: first draw and redo) and the largest number of late
// pieces in a node.  out4 = {operations, nodes, squeeze nodes, max pieces}.
int cg1_shuffle_fe_program_shape(size_t ell, size_t lg, uint32_t* out4) {
  if (!out4 || ell == 0 || lg == 0 || lg > 20 || ((ell + 4) != ((size_t)1 << lg))) return CG1_ERR_ARG;
  FeProgram P;
  fe_build_program(ell, lg, P);
  uint8_t init[CG1_MERLIN_STATE_BYTES], consts[96];
  cg1_merlin_init(init, (const uint8_t*)"curdleproofs", 12);
  memset(consts, 0, sizeof consts);
  consts[0] = 0xC0;
  std::vector<cg1fe::RowDesc> desc;
  uint32_t nn = 0;
  out4[0] = (uint32_t)P.ops.size(); out4[1] = out4[2] = out4[3] = 0;
  if (P.labels.size() > (size_t)cg1merlin::MAX_LABELS || !fe_build_nodes(P, init, consts, desc, nn)) return CG1_OK;
  out4[1] = nn;
  for (uint32_t nd = 0; nd < nn; ++nd) {
    const cg1fe::RowDesc* row = desc.data() + (size_t)nd * cg1fe::ROW_WORDS;
    if ((row[42].tword & 3u) == cg1fe::N_SQUEEZE) ++out4[2];
    uint32_t np = 0;
    for (uint32_t q = 0; q < cg1fe::MAX_PIECES; ++q) if (row[43 + q].tword & 63u) ++np;
    out4[3] = std::max(out4[3], np);
  }
  return CG1_OK;
}

// Nodes of the block program (0: the program of this ell does not fit the row format and the byte-machine kernel is used).
size_t cg1_shuffle_fe_nodes(const cg1_shuffle_fe* fe) { return fe ? fe->n_nodes : 0; }

// Keccak passes of the slowest wave of the last launch enqueued on `ctx` (waits for the stream; 0 if that launch used the byte machine).
size_t cg1_shuffle_fe_last_passes(cg1_shuffle_fe* fe, cg1_ctx* ctx) {
  if (!fe || !ctx || !fe->last_blocks || !fe->d_passes) return 0;
  if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return 0;
  std::vector<uint32_t> h(fe->last_blocks * 8);
  if (hipMemcpy(h.data(), fe->d_passes, h.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  size_t best = 0;
  for (size_t b = 0; b < fe->last_blocks; ++b) if (h[8 * b] > h[8 * best]) best = b;
  for (int k = 0; k < 8; ++k) fe->last_split[k] = h[8 * best + k];
  return h[8 * best];
}
// ... and that wave's shader-clock split (launches with cg1_ctx_set_param("fe_timed", 1)): out[0..7) = clocks / 256 spent in (late pieces +
// issuing the row loads, Keccak-f, whole passes, draw + range check, X_GPROD, X_DA, X_FINAL); valid after cg1_shuffle_fe_last_passes.
void cg1_shuffle_fe_last_split(const cg1_shuffle_fe* fe, uint32_t* out7) { for (int k = 0; k < 7; ++k) out7[k] = fe ? fe->last_split[k + 1] : 0; }

// Enqueue the front-end of n proofs on ctx's compute stream (no wait: cg1_stream_sync).  d_wire48: n x L own points as gathered from
// the wire (cg1_shuffle_gather_points); d_pts_affine96: the same points decoded (cg1_batch_decompress_*); d_aux: n x 19 x 32 bytes
// (cg1_shuffle_gather_aux); outputs as cg1_shuffle_prepare_inputs: d_rowin n x cg1_shuffle_rowin_scalars() x 32, d_status n codes.
int cg1_shuffle_fe_enqueue(cg1_shuffle_fe* fe, cg1_ctx* ctx, size_t n, const void* d_wire48, const void* d_pts_affine96, const void* d_aux,
                           void* d_rowin, void* d_status, int lanes_per_wave) {
  if (!fe || !ctx) return CG1_ERR_ARG;
  if (n == 0) return CG1_OK;
  if (!d_wire48 || !d_pts_affine96 || !d_aux || !d_rowin || !d_status || n >= (1u << 24) || ctx->device != fe->device) return CG1_ERR_ARG;
  if (lanes_per_wave < 1 || lanes_per_wave > cg1merlin::LANES) lanes_per_wave = cg1merlin::LANES;
  HIPCHK(hipSetDevice(ctx->device));
  if (n > fe->cap_n) {
    if (fe->d_four) (void)hipFree(fe->d_four);
    if (fe->d_scratch) (void)hipFree(fe->d_scratch);
    fe->d_four = fe->d_scratch = nullptr; fe->cap_n = 0;
    HIPCHK(hipMalloc(&fe->d_four, n * 4 * sizeof(cg1::PreparedPoint)));
    HIPCHK(hipMalloc(&fe->d_scratch, n * (size_t)fe->pr.out_stride));
    fe->cap_n = n;
  }
  hipLaunchKernelGGL(cg1fe::k_fe_gather4, dim3((unsigned)((4 * n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_pts_affine96, fe->pr, (uint32_t)n,
                     (cg1::PreparedPoint*)fe->d_four);
  const unsigned nblk = (unsigned)((n + lanes_per_wave - 1) / lanes_per_wave);
  fe->pr.prio = (uint32_t)ctx->fe_prio;
  if (fe->n_nodes && ctx->fe_rows) {
    const size_t row_words = (size_t)fe->n_nodes * cg1fe::ROW_WORDS;
    const size_t need = (size_t)nblk * lanes_per_wave * row_words;          // rows are laid out per wave: the last wave's unused lanes count
    if (need > fe->cap_rows || nblk > fe->cap_blocks) {
      if (fe->d_rows) (void)hipFree(fe->d_rows);
      if (fe->d_passes) (void)hipFree(fe->d_passes);
      fe->d_rows = fe->d_passes = nullptr; fe->cap_rows = 0; fe->cap_blocks = 0; fe->last_blocks = 0;
      HIPCHK(hipMalloc(&fe->d_rows, need * 4));
      HIPCHK(hipMalloc(&fe->d_passes, (size_t)nblk * 32));
      fe->cap_rows = need; fe->cap_blocks = nblk;
    }
    fe->last_blocks = nblk;
    hipLaunchKernelGGL(cg1merlin::k_fill_rows, dim3((unsigned)((row_words + 255) / 256), (unsigned)std::min<size_t>(n, 65535)), dim3(256), 0, ctx->stream, (const cg1fe::RowDesc*)fe->d_desc, fe->n_nodes,
                       (const uint8_t*)d_wire48, (size_t)fe->pr.L * 48, 1u, (uint32_t)n, (uint32_t)lanes_per_wave, (uint32_t*)fe->d_rows);
    hipLaunchKernelGGL(ctx->fe_timed ? cg1fe::k_shuffle_front_end_rows<true> : cg1fe::k_shuffle_front_end_rows<false>, dim3(nblk), dim3(cg1merlin::LANES), 0, ctx->stream, (const uint8_t*)fe->d_init, (const uint32_t*)fe->d_rows,
                       fe->n_nodes, (const uint8_t*)d_wire48, (const uint8_t*)d_aux, (const cg1::PreparedPoint*)fe->d_four, (const cg1::PreparedPoint*)fe->d_tabG,
                       (const cg1::PreparedPoint*)fe->d_tabH, fe->pr, (uint8_t*)fe->d_scratch, (uint8_t*)d_rowin, (int32_t*)d_status, (uint32_t)n,
                       (uint32_t)lanes_per_wave, (uint32_t*)fe->d_passes);
    HIPCHK(hipGetLastError());
    return CG1_OK;
  }
  fe->last_blocks = 0;
  hipLaunchKernelGGL(cg1fe::k_shuffle_front_end, dim3(nblk), dim3(cg1merlin::LANES), 0, ctx->stream, (const uint8_t*)fe->d_init, (const cg1merlin::COp*)fe->d_ops,
                     fe->nops, (const uint32_t*)fe->d_labels, fe->nlabels, (const uint8_t*)fe->d_consts, (const uint8_t*)d_wire48, (const uint8_t*)d_aux,
                     (const cg1::PreparedPoint*)fe->d_four, (const cg1::PreparedPoint*)fe->d_tabG, (const cg1::PreparedPoint*)fe->d_tabH, fe->pr,
                     (uint8_t*)fe->d_scratch, (uint8_t*)d_rowin, (int32_t*)d_status, (uint32_t)n, (uint32_t)lanes_per_wave);
  HIPCHK(hipGetLastError());
  return CG1_OK;
}

}  // extern "C"

extern "C" {

int cg1_shuffle_rows_device(cg1_ctx* ctx, size_t ell, size_t lg, size_t n_proofs, const void* d_rowin, const void* d_host_status,
                            const void* d_point_status, void* d_out_scalars, void* d_crs_rows, void* d_status_out) {
  if (!ctx) return CG1_ERR_HIP;
  if (n_proofs == 0) return CG1_OK;
  if (!d_rowin || !d_host_status || !d_point_status || !d_out_scalars || !d_crs_rows || !d_status_out || lg >= 32 || ell + 4 != ((size_t)1 << lg))
    return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t L = 4 * ell + 19 + 10 * lg, C = ell + 9;
  hipLaunchKernelGGL(cg1rows::k_shuffle_rows, dim3((unsigned)n_proofs), dim3(128), 0, ctx->stream, (const uint8_t*)d_rowin,
                     (const int32_t*)d_host_status, (const uint8_t*)d_point_status, (uint32_t)ell, (uint32_t)lg,
                     (uint8_t*)d_out_scalars, (uint8_t*)d_crs_rows, (int32_t*)d_status_out);
  hipLaunchKernelGGL(cg1rows::k_crs_row_sum, dim3((unsigned)C), dim3(256), 0, ctx->stream, (const uint8_t*)d_crs_rows,
                     (const int32_t*)d_status_out, (uint32_t)n_proofs, (uint32_t)C, (uint8_t*)d_out_scalars + n_proofs * L * 32);
  HIPCHK(hipGetLastError());
  return CG1_OK;
}
int cg1_batch_compress_device(cg1_ctx* ctx, const void* d_in_affine96, void* d_out48, size_t n) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) return CG1_OK;
  if (n >= (1ull << 31)) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(cg1::k_batch_compress, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const uint32_t*)d_in_affine96, (uint8_t*)d_out48, (uint32_t)n);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  return CG1_OK;
}
// host buffers: returns CG1_OK if every encoding is valid, else the first failing status with *bad_index set
int cg1_batch_decompress_gpu(cg1_ctx* ctx, const uint8_t* in48, uint8_t* out_affine96, size_t n, int check_subgroup, size_t* bad_index) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) return CG1_OK;
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf din, dout, dst;
  HIPCHK(din.alloc(n * 48)); HIPCHK(dout.alloc(n * 96)); HIPCHK(dst.alloc(n));
  HIPCHK(hipMemcpy(din.p, in48, n * 48, hipMemcpyHostToDevice));
  int rc = cg1_batch_decompress_device(ctx, din.p, dout.p, dst.p, n, check_subgroup);
  if (rc != CG1_OK) return rc;
  std::vector<uint8_t> st(n);
  HIPCHK(hipMemcpy(out_affine96, dout.p, n * 96, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(st.data(), dst.p, n, hipMemcpyDeviceToHost));
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
// chip-wide v_mad_u64_u32 rate (lane-operations per second) at `waves_per_simd` resident waves, hipEvents on the context's stream
int cg1_probe_mad_rate(cg1_ctx* ctx, int waves_per_simd, int iters, double* lane_ops_per_s) {
  if (!ctx) return CG1_ERR_HIP;
  if (waves_per_simd < 1 || waves_per_simd > 8 || iters < 1 || !lane_ops_per_s) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, ctx->device));
  const unsigned blocks = (unsigned)prop.multiProcessorCount * (unsigned)waves_per_simd;       // 256 threads = 4 waves = one per SIMD of a CU
  DevBuf out;
  HIPCHK(out.alloc(256));
  hipLaunchKernelGGL(cg1::k_probe_mad_rate, dim3(blocks), dim3(256), 0, ctx->stream, (uint32_t*)out.p, 8, 12345u);      // warm-up
  HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
  hipLaunchKernelGGL(cg1::k_probe_mad_rate, dim3(blocks), dim3(256), 0, ctx->stream, (uint32_t*)out.p, iters, 12345u);
  HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
  *lane_ops_per_s = (double)blocks * 256.0 * (double)iters * 128.0 / ((double)ms * 1e-3);
  return CG1_OK;
}

// out[j] = sum of points [offsets[j], offsets[j+1]) (affine96 in and out; offsets: HOST array of n_groups + 1 entries)
int cg1_batch_sum_device(cg1_ctx* ctx, const void* d_points, const uint32_t* offsets, size_t n_groups, void* d_out) {
  if (!ctx) return CG1_ERR_HIP;
  if (n_groups == 0) return CG1_OK;
  if (!d_points || !offsets || !d_out || offsets[0] != 0 || n_groups >= (1u << 30)) return CG1_ERR_ARG;
  for (size_t j = 0; j < n_groups; ++j) if (offsets[j + 1] < offsets[j]) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf offs;
  HIPCHK(offs.alloc((n_groups + 1) * 4));
  HIPCHK(hipMemcpyAsync(offs.p, offsets, (n_groups + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(cg1::k_batch_sum, dim3((unsigned)n_groups), dim3(64), 0, ctx->stream, (const uint32_t*)d_points, (const uint32_t*)offs.p, (uint32_t*)d_out);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  return CG1_OK;
}
int cg1_batch_sum(cg1_ctx* ctx, const uint8_t* points, const uint32_t* offsets, size_t n_groups, uint8_t* out) {
  if (!ctx) return CG1_ERR_HIP;
  if (n_groups == 0) return CG1_OK;
  if (!points || !offsets || !out) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t n = offsets[n_groups];
  DevBuf in, res;
  HIPCHK(in.alloc(96 * (n ? n : 1)));
  HIPCHK(res.alloc(96 * n_groups));
  if (n) HIPCHK(hipMemcpy(in.p, points, 96 * n, hipMemcpyHostToDevice));
  int rc = cg1_batch_sum_device(ctx, in.p, offsets, n_groups, res.p);
  if (rc) return rc;
  HIPCHK(hipMemcpy(out, res.p, 96 * n_groups, hipMemcpyDeviceToHost));
  return CG1_OK;
}

int cg1_probe_madd(cg1_ctx* ctx, const void* d_points, size_t npts, size_t lanes, int iters, float* ms) {
  if (!ctx) return CG1_ERR_HIP;
  if (npts == 0 || lanes == 0 || lanes % 256) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf prep, out, fl;
  HIPCHK(prep.alloc(npts * sizeof(cg1::PreparedPoint)));
  HIPCHK(fl.alloc(npts + 16));
  HIPCHK(out.alloc(lanes * sizeof(cg1::PointSum)));
  hipLaunchKernelGGL(cg1::k_prepare_points, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, ctx->stream, (const uint32_t*)d_points,
                     (cg1::PreparedPoint*)prep.p, (uint8_t*)fl.p, (uint32_t)npts);
  hipLaunchKernelGGL(cg1::k_probe_madd, dim3((unsigned)(lanes / 256)), dim3(256), 0, ctx->stream, (cg1::PreparedPoint*)prep.p, (uint32_t)npts, (cg1::PointSum*)out.p, 2);
  HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
  hipLaunchKernelGGL(cg1::k_probe_madd, dim3((unsigned)(lanes / 256)), dim3(256), 0, ctx->stream, (cg1::PreparedPoint*)prep.p, (uint32_t)npts, (cg1::PointSum*)out.p, iters);
  HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventElapsedTime(ms, ctx->ev[0], ctx->ev[1]));
  return CG1_OK;
}

// `waves` waves each run `iters` dependent EC additions (k_probe_add_chain, fp_row.h): mode 0 = one lane per addition, 1 = a DPP quad,
// 2 = one limb per lane.  *ms: device time of one launch (hipEvents, best of `reps`); out_blob: wave 0's result.
int cg1_probe_add_chain(cg1_ctx* ctx, int mode, const uint8_t* two_points_affine96, size_t waves, int iters, int reps, uint8_t* out_blob, float* ms) {
  if (!ctx) return CG1_ERR_HIP;
  if (mode < 0 || mode > 2 || !two_points_affine96 || waves == 0 || waves > (1u << 20) || iters < 0 || reps < 1 || !ms) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf raw, prep, fl, out;
  HIPCHK(raw.alloc(2 * 96)); HIPCHK(prep.alloc(2 * sizeof(cg1::PreparedPoint))); HIPCHK(fl.alloc(32)); HIPCHK(out.alloc(waves * sizeof(cg1::PointWords)));
  HIPCHK(hipMemcpy(raw.p, two_points_affine96, 2 * 96, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(cg1::k_prepare_points, dim3(1), dim3(256), 0, ctx->stream, (const uint32_t*)raw.p, (cg1::PreparedPoint*)prep.p, (uint8_t*)fl.p, 2u, (uint32_t*)nullptr);
  auto launch = [&](int it) {
    const dim3 g((unsigned)waves), b(64);
    if (mode == 0) hipLaunchKernelGGL((cg1::k_probe_add_chain<0>), g, b, 0, ctx->stream, (const cg1::PreparedPoint*)prep.p, (cg1::PointWords*)out.p, it);
    else if (mode == 1) hipLaunchKernelGGL((cg1::k_probe_add_chain<1>), g, b, 0, ctx->stream, (const cg1::PreparedPoint*)prep.p, (cg1::PointWords*)out.p, it);
    else hipLaunchKernelGGL((cg1::k_probe_add_chain<2>), g, b, 0, ctx->stream, (const cg1::PreparedPoint*)prep.p, (cg1::PointWords*)out.p, it);
  };
  launch(2);                                             // warm-up (code object load, instruction cache)
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    HIPCHK(hipEventRecord(ctx->ev[0], ctx->stream));
    launch(iters);
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipGetLastError());
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, ctx->ev[0], ctx->ev[1]));
    if (t < best) best = t;
  }
  *ms = best;
  if (out_blob) {
    cg1::PointWords w;
    HIPCHK(hipMemcpy(&w, out.p, sizeof w, hipMemcpyDeviceToHost));
    blob_out(out_blob, cg1::jac_from_words(w));
  }
  return CG1_OK;
}

}  // extern "C"
