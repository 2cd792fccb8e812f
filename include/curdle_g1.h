/* curdle_g1.h -- C ABI of libcurdle_g1.so: the MI355X-native BLS12-381 G1 / MSM engine that drops in
 * behind curdleproofs.pie's compute_MSM / MSMAccumulator and the G1Point surface of py_arkworks_bls12381.
 *
 * The reference has no FFI of its own: its boundary is the Python import
 *     from py_arkworks_bls12381 import G1Point, Scalar
 * (curdleproofs/curdleproofs/util.py:4, msm_accumulator.py:3, ...) plus
 *     from curdleproofs.msm_accumulator import MSMAccumulator, compute_MSM
 * (curdleproofs.py:17, ipa.py:23, grand_prod.py:17, same_msm.py:19, same_perm.py:14).
 * Each entry point below names the reference interface it stands behind.  The ctypes binding a
 * maintainer adds is shown in INTEGRATION.md (and lives in curdleproofs_pie_amd/_native.py).
 *
 * Conventions
 *   - plain pointers and sizes only; every function returns an int status (CG1_OK == 0) unless it
 *     cannot fail; no exceptions cross the boundary; no global mutable state outside a cg1_ctx.
 *   - "point blob": 144 opaque bytes (host Jacobian X,Y,Z; 6x64-bit Montgomery limbs each).
 *   - "affine96": x || y, each a 48-byte little-endian integer < p in standard (non-Montgomery) form;
 *     the all-zero record encodes the identity ((0,0) is not on the curve).
 *   - "scalar32": 32-byte little-endian canonical Fr element (< r), as Scalar.to_le_bytes() returns.  The MSM entry
 *     points treat it as a plain integer and reject values >= 2^255 (CG1_ERR_ENCODING) instead of mis-summing them.
 *   - "compressed48": the 48-byte ZCash-format compression G1Point.to_compressed_bytes() returns.
 *   - device entry points fail with CG1_ERR_HIP when no GPU / HIP error; there is NO CPU fallback.
 */
#ifndef CURDLE_G1_H
#define CURDLE_G1_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CG1_OK            0
#define CG1_ERR_ARG       1   /* bad argument (size, window width, shard spec) */
#define CG1_ERR_HIP       2   /* HIP runtime error or no device; message via cg1_ctx_error */
#define CG1_ERR_ENCODING  3   /* malformed compressed48 / affine96 / scalar32 -> Python ValueError */
#define CG1_ERR_NOT_ON_CURVE 4
#define CG1_ERR_NOT_IN_SUBGROUP 5
#define CG1_ERR_COMM      6   /* multi-GPU exchange failed (rendezvous, socket, RCCL); message via cg1_comm_error */

#define CG1_POINT_BYTES 144
#define CG1_NPHASE 7          /* prepare, sort_count, sort_scatter, chunks, accumulate, seg_reduce, bit_tree(+D2H) */

typedef struct cg1_ctx cg1_ctx;

/* ---------------- host-side single-element G1 ops (G1Point operators; stub __init__.pyi:5-30) ---- */
void cg1_identity(uint8_t out[CG1_POINT_BYTES]);                      /* G1Point.identity()        util.py:11 */
void cg1_generator(uint8_t out[CG1_POINT_BYTES]);                     /* G1Point()                 util.py:9  */
void cg1_add(uint8_t out[CG1_POINT_BYTES], const uint8_t* a, const uint8_t* b);      /* __add__ */
void cg1_sub(uint8_t out[CG1_POINT_BYTES], const uint8_t* a, const uint8_t* b);      /* __sub__ */
void cg1_neg(uint8_t out[CG1_POINT_BYTES], const uint8_t* a);                        /* __neg__ */
void cg1_double(uint8_t out[CG1_POINT_BYTES], const uint8_t* a);
void cg1_mul(uint8_t out[CG1_POINT_BYTES], const uint8_t* a, const uint8_t scalar32[32]);  /* __mul__(Scalar) */
int  cg1_eq(const uint8_t* a, const uint8_t* b);                      /* __eq__; 1 equal, 0 not    util.py:17-18 */
int  cg1_is_identity(const uint8_t* a);
void cg1_compress(uint8_t out48[48], const uint8_t* a);               /* to_compressed_bytes       util.py:27-28 */
/* from_compressed_bytes (check_subgroup=1) / from_compressed_bytes_unchecked (0)   util.py:35-36,
 * msm_accumulator.py:65.  Returns CG1_OK or CG1_ERR_ENCODING / _NOT_ON_CURVE / _NOT_IN_SUBGROUP. */
int  cg1_decompress(uint8_t out[CG1_POINT_BYTES], const uint8_t in48[48], int check_subgroup);
void cg1_to_affine96(uint8_t out96[96], const uint8_t* a);
int  cg1_from_affine96(uint8_t out[CG1_POINT_BYTES], const uint8_t in96[96], int check_on_curve);
/* n affine96 records (as the batched device entry points return them; zeros = identity; not checked against the curve) -> n blobs */
int  cg1_batch_from_affine96(uint8_t* out_blobs, const uint8_t* in96, size_t n);
/* n point blobs -> n affine96 records with ONE field inversion (input marshalling for the MSM) */
void cg1_batch_to_affine96(uint8_t* out96, const uint8_t* blobs, size_t n);
/* n compressed48 -> n point blobs; stops at the first bad encoding and returns its status, *bad_index set */
int  cg1_batch_decompress(uint8_t* out_blobs, const uint8_t* in48, size_t n, int check_subgroup, size_t* bad_index);
void cg1_batch_compress(uint8_t* out48, const uint8_t* blobs, size_t n);

/* ---------------- device context --------------------------------------------------------------- */
int  cg1_device_count(void);                                         /* 0 when no GPU is visible */
cg1_ctx* cg1_ctx_create(int device);                                 /* NULL on failure (no GPU) */
/* the same with the context's compute / side streams confined to the CUs whose bit is set (bit i of word i / 32 = CU i) */
cg1_ctx* cg1_ctx_create_cu_mask(int device, const uint32_t* cu_mask, size_t n_words);
void cg1_ctx_destroy(cg1_ctx* ctx);
const char* cg1_ctx_error(const cg1_ctx* ctx);                       /* message of the last failure */
void* cg1_dev_malloc(cg1_ctx* ctx, size_t bytes);                    /* NULL on failure */
void cg1_dev_free(cg1_ctx* ctx, void* p);
int  cg1_h2d(cg1_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int  cg1_d2h(cg1_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* asynchronous H2D on the context's copy stream (src page-locked to overlap kernels); cg1_copy_fence orders everything
 * queued there so far before the next launch on the compute stream.  Neither blocks the host. */
int  cg1_h2d_async(cg1_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int  cg1_copy_fence(cg1_ctx* ctx);
int  cg1_stream_sync(cg1_ctx* ctx);                                  /* this context's compute stream only */
/* page-locked host memory for staging buffers (copies from it run at full PCIe rate); NULL on failure */
void* cg1_host_alloc(cg1_ctx* ctx, size_t bytes);
void cg1_host_free(cg1_ctx* ctx, void* p);
/* strided gather: `rows` records of `width` bytes lying `src_pitch` apart on the device -> `dst_pitch` apart on the host */
int  cg1_d2h_2d(cg1_ctx* ctx, void* dst_host, size_t dst_pitch, const void* src_dev, size_t src_pitch, size_t width, size_t rows);
int  cg1_ctx_sync(cg1_ctx* ctx);                                     /* hipDeviceSynchronize on the context's GPU */
int  cg1_ctx_device(const cg1_ctx* ctx);                             /* the HIP device ordinal the context was created on */
void* cg1_ctx_stream(cg1_ctx* ctx);                                  /* the context's compute stream (a hipStream_t), for callers ordering their own work */
/* Tuning and A/B switches of a context (defaults are the measured best; DESIGN.md section 9 has the measurements):
 *   MSM plan / phases   "chunk_len" "seg_m" "auto_plan" "stage_sort" "partition_sort" "big_bins" "wave_agg" "quad" "reduce_2d" "rowcol_quad"
 *                       "rowcol_quad_max" "fold_pass" "tree_half" "scan_one" "host_split" "horner_threads" "zero_copy" "batched_host_horner_max"
 *                       "batch_mul_quad_max" "batch_mul_host_max" "batch_mul_row" "horner_row" "small_row_tail" "sort_sub_bits" "rowcol_lgq" "tree_shift" "arm_helpers" "small_msm" (1: calls of <= 2048 terms run as ONE launch, k_msm_small; 0: the regime-A chain)
 *                       "lincomb_zero_copy" (1: GPU shares of cg1_lincomb_batch up to 4 096 terms are read by the kernel from mapped host memory, no staged copy),
 *                       "fold_quad" (1: quads fold a bucket's chunk sums at small bucket counts; 0: one lane per bucket, measured slower),
 *                       "tree_row" / "rowcol_row" (1: regime A's item sums, and for <= 2^18 buckets the cross-quad levels of the row / column sums, with one
 *                       limb per lane; 0: the quad kernels), "glv" / "glv_max_n" (the endomorphism split, see below: a PROMISE about the points, not a tuning
 *                       knob), "batched_split" / "batched_split_min_m" (regime B as two staggered half-batches: measured slower, default 0),
 *                       "split" (A/B switch, default 0: a call of >= 2^"split_min_log2n" terms as two launch chains -- high and low half of its windows -- on
 *                       two streams; measured slower than the single chain, profiles/r04_split_ab.txt)
 *   waiting             "blocking_sync" (sleep instead of spinning on the stream), "profile" (0: no events, 1: around k_accumulate, 2: every phase)
 *   codec               "decompress_waves" (2 | 3: waves per SIMD k_batch_decompress is compiled for)
 *   transcripts         "merlin_rows" (1: block program when the operation list fits), "merlin_sync" (1: lanes of a wave permute together),
 *                       "merlin_lanes" (1..64 transcripts per wave)
 *   verifier front-end  "fe_rows" (1: block program, 0: byte-level machine), "fe_timed" (shader-clock split of a launch), "fe_prio" (0..3)
 * Unknown names and out-of-range values return CG1_ERR_ARG. */
int  cg1_ctx_set_param(cg1_ctx* ctx, const char* name, int value);

/* ---------------- the hot path: compute_MSM  (msm_accumulator.py:6-12) ------------------------- */
/* sum_i scalars[i] * points[i], inputs in host memory (copied to the device by the call). */
int cg1_msm(cg1_ctx* ctx, const uint8_t* points_affine96, const uint8_t* scalars32, size_t n,
            uint8_t out[CG1_POINT_BYTES]);
/* Same with inputs already resident in device memory (hipMalloc / cg1_dev_malloc / a torch tensor's
 * data_ptr).  window_c: 4..16 = uniform signed-digit windows of that width; -16..-4 = the balanced plan with cmax = -window_c
 * (ceil(256 / cmax) windows of width cmax or cmax - 1, so no window is thin); 0 = choose from n.
 * (shard_rank, shard_world): this call sums only windows w = shard_rank (mod shard_world) and returns
 * sum_w 2^(c w) S_w for those -- the per-GPU partial of a window-sharded MSM; (0,1) = the whole MSM.
 * window_c must then be the same on every rank. */
int cg1_msm_device(cg1_ctx* ctx, const void* d_points_affine96, const void* d_scalars32, size_t n,
                   int window_c, int shard_rank, int shard_world, uint8_t out[CG1_POINT_BYTES]);
/* Regime B -- a batch of n_msm INDEPENDENT small MSMs in one launch chain (e.g. the 5*ell+7-term final MSMs of
 * many MSMAccumulator.verify() calls, msm_accumulator.py:60-68; BASELINE config 3: 1024 x 627 terms).
 * MSM j sums terms [offsets[j], offsets[j+1]) of the concatenated inputs; offsets is a HOST array of
 * n_msm+1 entries with offsets[0] == 0; out_blobs receives n_msm point blobs.  window_c 4..9, 0 = auto. */
int cg1_msm_batched_device(cg1_ctx* ctx, const void* d_points_affine96, const void* d_scalars32,
                           const uint32_t* offsets, size_t n_msm, int window_c, uint8_t* out_blobs);
/* cg1_msm_device in two halves: _begin enqueues the context's launch chain and returns, _end waits, runs the host tail and
 * delivers the point (identity when n was 0).  Two contexts on one GPU keep two MSMs in flight: the next one's sort phases
 * run under this one's bucket accumulation, this one's reduction tail under the next one's. */
int cg1_msm_device_begin(cg1_ctx* ctx, const void* d_points_affine96, const void* d_scalars32, size_t n, int window_c,
                         int shard_rank, int shard_world);
int cg1_msm_device_end(cg1_ctx* ctx, uint8_t out[CG1_POINT_BYTES]);
int cg1_msm_batched(cg1_ctx* ctx, const uint8_t* points_affine96, const uint8_t* scalars32,
                    const uint32_t* offsets, size_t n_msm, uint8_t* out_blobs);
/* ---- the same call as the reference's callers make it: over the G1Point OBJECTS' own blobs (msm_accumulator.py:6-12 takes lists
 * of G1Point / Scalar; a G1Point here holds a 144-byte point blob, generally not normalised).  The blobs are uploaded as they are
 * and normalised on the device (k_prepare_blobs: Montgomery's trick, one inversion per GPU lane), so the host does no field
 * arithmetic per call.  all_normalised != 0: every blob has Z = 1 or Z = 0 (the caller checked): no inversion at all. */
int cg1_msm_blobs(cg1_ctx* ctx, const uint8_t* blobs144, const uint8_t* scalars32, size_t n, int all_normalised,
                  uint8_t out[CG1_POINT_BYTES]);
/* The same in two halves, for a caller that uploads slice by slice while it is still gathering the next one (cg1_h2d_async +
 * cg1_copy_fence): cg1_stage_reserve hands out the context's device staging for n blobs / n scalars (valid until the next call that
 * stages more), cg1_msm_blobs_device sums over blobs already resident there (or anywhere else on the device). */
int cg1_stage_reserve(cg1_ctx* ctx, size_t pts_bytes, size_t sc_bytes, void** d_pts, void** d_sc);
int cg1_msm_blobs_device(cg1_ctx* ctx, const void* d_blobs144, const void* d_scalars32, size_t n, int all_normalised,
                         uint8_t out[CG1_POINT_BYTES]);
/* A point vector resident on the device in the accumulation kernels' record format -- crs.vec_G / vec_H / vec_R ... are the
 * bases of dozens of compute_MSM calls of one prover (curdleproofs.py:77,94,95,319; grand_prod.py:54,90; ipa.py:97,98): made
 * once (upload + k_prepare_blobs), then each MSM uploads only its scalars.  cg1_msm_vec sums scalars[i] * vec[first + i], i < n. */
typedef struct cg1_vec cg1_vec;
cg1_vec* cg1_vec_create(cg1_ctx* ctx, const uint8_t* blobs144, size_t n, int all_normalised);   /* NULL on failure */
void   cg1_vec_destroy(cg1_vec* v);
size_t cg1_vec_len(const cg1_vec* v);
int cg1_msm_vec(cg1_ctx* ctx, const cg1_vec* vec, size_t first, size_t n, const uint8_t* scalars32, uint8_t out[CG1_POINT_BYTES]);
/* Host: n point blobs -> affine96 and / or compressed48 (either may be NULL) with ONE shared inversion: the map key
 * (msm_accumulator.py:54) and the affine form of MSMAccumulator.accumulate_check's bases from a single normalisation. */
int cg1_batch_normalize(const uint8_t* blobs, size_t n, uint8_t* out_affine96, uint8_t* out_comp48);

/* per-phase GPU times (hipEvents on the context's stream) and host Horner tail of the last MSM call */
int cg1_get_timings(const cg1_ctx* ctx, float phase_ms[CG1_NPHASE], float* host_tail_ms, int* window_c);

/* work counts of the last MSM call: non-zero signed digits sorted into buckets (= bucket additions + first-entry copies)
 * and the chunks k_accumulate ran (one copy each), so  mixed additions = entries - chunks */
int cg1_get_last_counts(const cg1_ctx* ctx, uint32_t* entries, uint32_t* chunks);
/* k_accumulate launches of the last MSM call: 2 when it ran as two launch chains ("split"), 1, or 0 when k_msm_small served it */
int cg1_get_last_launches(const cg1_ctx* ctx);
/* The window plan of a width (window_c > 0: uniform; < 0: balanced with widths |c| and |c| - 1; glv: over the 128 positions of the halves of
 * the endomorphism split instead of 256): offset and width of every window, low to high.  Returns the number of windows (<= capacity) or -1.
 * No GPU involved: the CPU tests check that every plan tiles the bit positions from 0 without a gap. */
int cg1_plan_describe(int window_c, int glv, int* out_offsets, int* out_widths, int capacity);
/* hipEvent stopwatch on the context's compute stream: device time of everything enqueued between begin and end */
int cg1_timer_begin(cg1_ctx* ctx);
int cg1_timer_end(cg1_ctx* ctx, float* ms);

/* host-side wall times of the last MSM call: enqueue, wait-for-GPU, event readout, Horner tail (ms) */
int cg1_get_host_timings(const cg1_ctx* ctx, float host_ms[4]);

/* Segmented point sum -- the linear point sum of crs.py:64-65 (G_sum = reduce(a + b, vec_G, Z1), H_sum): out[j] = sum of the
 * points [offsets[j], offsets[j+1]); affine96 in and out; offsets is a HOST array of n_groups + 1 entries, offsets[0] == 0.
 * One wave per group. */
int cg1_batch_sum_device(cg1_ctx* ctx, const void* d_points_affine96, const uint32_t* offsets, size_t n_groups, void* d_out_affine96);
int cg1_batch_sum(cg1_ctx* ctx, const uint8_t* points_affine96, const uint32_t* offsets, size_t n_groups, uint8_t* out_affine96);
/* chip-wide v_mad_u64_u32 issue rate (lane-operations/s) with `waves_per_simd` resident waves: the peak roofline_int_mad is
 * priced against, measured in the same run as the benchmark (bench.py). */
int cg1_probe_mad_rate(cg1_ctx* ctx, int waves_per_simd, int iters, double* lane_ops_per_s);

/* ---------------- multi-GPU (SURVEY.md 8(b) "multi-GPU variants taking a device list", 8(e)) ---- */
/* The reference has no multi-device code; the contract is BASELINE.json's north_star: window buckets sharded over
 * the GPUs of one node, "final RCCL all-reduce of partial G1 sums".  RCCL has no elliptic-curve reduction, so the all-reduce is an
 * all-gather of ONE 144-byte point blob per rank + world-1 host additions on every rank (bit-identical on all ranks).
 *
 * (1) One process, several GPUs: ctxs[i] owns point shard i, resident on ITS device.  All launch chains are enqueued before
 *     any is waited for; out = the sum of the partials = compute_MSM (msm_accumulator.py:6-12) over the concatenation. */
int cg1_msm_multi_device(cg1_ctx* const* ctxs, size_t n_ctx, const void* const* d_points_affine96, const void* const* d_scalars32,
                         const size_t* n, int window_c, uint8_t out[CG1_POINT_BYTES]);
/* (2) One process per GPU (bench.py --gpus N, distributed.py).  A communicator is a TCP control channel on the loopback
 *     interface (rank 0 = hub: rendezvous, barriers, clocks, verdict gathers; alone it is the whole exchange when ranks rehearse on
 *     one GPU or on a CPU-only box) to which RCCL is attached for the data exchange on a GPU node.  No PyTorch anywhere.
 *       rank 0:  c = cg1_comm_create(0, world); publish cg1_comm_port(c) (file / env); cg1_comm_connect(c, NULL, 0, nonce, ms)
 *       others:  c = cg1_comm_create(r, world); cg1_comm_connect(c, "127.0.0.1", port, nonce, ms)   (CG1_ERR_COMM = wrong or
 *                not-yet-there listener: re-read the rendezvous and call again)
 *       all:     cg1_comm_attach_rccl(c, ctx)   -- ncclGetUniqueId on rank 0, id broadcast over the control channel,
 *                ncclCommInitRank on ctx's device; librccl.so is dlopen'ed here, on first use. */
typedef struct cg1_comm cg1_comm;
cg1_comm* cg1_comm_create(int rank, int world);                        /* NULL: bad arguments / cannot listen */
int  cg1_comm_port(const cg1_comm* c);                                 /* rank 0: the loopback port it listens on */
int  cg1_comm_rank(const cg1_comm* c);
int  cg1_comm_connect(cg1_comm* c, const char* host_ipv4, int port, uint64_t nonce, int timeout_ms);
int  cg1_comm_set_timeout(cg1_comm* c, int timeout_ms);                /* per collective; default 120 s */
int  cg1_comm_attach_rccl(cg1_comm* c, cg1_ctx* ctx);                  /* collective over all ranks */
const char* cg1_comm_transport(const cg1_comm* c);                     /* "rccl" once attached, else "socket" */
int  cg1_comm_world_seen(const cg1_comm* c);                           /* RCCL attached: ncclCommCount; else connected ranks */
const char* cg1_comm_error(const cg1_comm* c);
/* every rank contributes `bytes` host bytes, every rank receives world * bytes in rank order.  RCCL attached: one
 * ncclAllGather(uint8) on ctx's compute stream (payload staged through pinned memory); else the TCP star. */
int  cg1_comm_allgather(cg1_comm* c, const void* send, size_t bytes, void* recv);
int  cg1_comm_allgather_host(cg1_comm* c, const void* send, size_t bytes, void* recv);   /* always the control channel */
int  cg1_comm_barrier(cg1_comm* c);                                    /* control channel; pair with cg1_ctx_sync */
/* the "all-reduce of partial G1 sums": sum = partial_0 + ... + partial_{world-1}; all_blobs (may be NULL) gets the world blobs */
int  cg1_comm_allreduce_g1(cg1_comm* c, const uint8_t partial[CG1_POINT_BYTES], uint8_t sum[CG1_POINT_BYTES], uint8_t* all_blobs);
void cg1_comm_destroy(cg1_comm* c);

/* ---------------- batched scalar multiplication (`G1Point * Scalar`, vectorised) --------------- */
/* out[i] = scalars[i] * bases[i % nbase]; all device pointers; affine96 in and out.
 * nbase = 1 is the fixed-base case (get_random_point: G * random_scalar(), util.py:67-68). */
int cg1_batch_mul_device(cg1_ctx* ctx, const void* d_bases_affine96, size_t nbase, const void* d_scalars32,
                         void* d_out_affine96, size_t n);
/* General form: out[i] = addend[i] + scalars[i % nscalars] * bases[i % nbase]   (addend may be NULL).
 * nscalars = 1: same-scalar map  [R*k for R in vec_R]  (curdleproofs.py:310-311);
 * nscalars = 1 with addend = L:  fold  G_L[i] + G_R[i]*gamma  (ipa.py:142-146, same_msm.py:122-126);
 * per-index scalars:             G_i * beta^-i  (grand_prod.py:64-71). */
int cg1_batch_mul_add_device(cg1_ctx* ctx, const void* d_bases_affine96, size_t nbase, const void* d_scalars32,
                             size_t nscalars, const void* d_addend_affine96, void* d_out_affine96, size_t n);
/* same, all buffers in host memory (copied in and out by the call).  Which engine serves a call depends on the call alone ("batch_mul_host_max":
 * -1 = this rule, 0 = never the host, N = the host up to N outputs): up to 96 outputs the host's worker pool (cg1_batch_mul_add_pool: n scalar
 * multiplications of ~77 us over its threads, against a ~0.6 ms launch); up to 4 096 one WAVE per output with one limb per lane (k_batch_mul_row,
 * csrc/fp_row.h: ~0.55 ms whatever n is; "batch_mul_row" = 0 switches it off); beyond, one quad / one lane per output.  Every engine returns the
 * same records, byte for byte, and refuses the same inputs: a coordinate >= p is CG1_ERR_ENCODING; the curve equation is not checked. */
int cg1_batch_mul_add_pool(const uint8_t* bases_affine96, size_t nbase, const uint8_t* scalars32, size_t nscalars,
                           const uint8_t* addend_affine96, uint8_t* out_affine96, size_t n, int n_threads /* 0 = all */);
int cg1_batch_mul_add(cg1_ctx* ctx, const uint8_t* bases_affine96, size_t nbase, const uint8_t* scalars32, size_t nscalars,
                      const uint8_t* addend_affine96, uint8_t* out_affine96, size_t n);
/* Batched 48-byte decompression on the GPU: from_compressed_bytes_unchecked
 * (check_subgroup = 0, util.py:35-36, BufReader.read_g1 util.py:143-147) / from_compressed_bytes (= 1).
 * Device variant: per-point status byte (0 ok, CG1_ERR_ENCODING, _NOT_ON_CURVE, _NOT_IN_SUBGROUP), affine96 out.
 * Host variant: CG1_OK iff all n encodings are valid, else the first failing status and *bad_index. */
int cg1_batch_decompress_device(cg1_ctx* ctx, const void* d_in48, void* d_out_affine96, void* d_status, size_t n, int check_subgroup);
int cg1_batch_decompress_enqueue(cg1_ctx* ctx, const void* d_in48, void* d_out_affine96, void* d_status, size_t n, int check_subgroup);  /* no wait: pair with cg1_ctx_sync */
int cg1_batch_decompress_gpu(cg1_ctx* ctx, const uint8_t* in48, uint8_t* out_affine96, size_t n, int check_subgroup, size_t* bad_index);
/* Subgroup flags for k selected points of every proof of a decompressed batch (d_affine96: n_proofs x stride_points
 * records; offsets[j] < stride_points, k <= 16): d_flags[proof * k + j] = 1 iff that point is on the curve but outside G1.
 * The reference decodes unchecked (util.py:35-36) yet asserts the same-scalar equalities exactly (same_scalar.py:108);
 * weighting them randomly is sound only for points of G1, so flagged proofs get the exact check below.  Runs on the
 * context's side stream, ordered after what the compute stream holds so far; cg1_side_sync waits for it. */
int cg1_subgroup_flags_enqueue(cg1_ctx* ctx, const void* d_affine96, size_t stride_points, size_t n_proofs,
                               const uint32_t* offsets, size_t k, void* d_flags);
int cg1_side_sync(cg1_ctx* ctx);
/* Batched compression on the GPU: n affine96 records (as cg1_batch_mul_add_device / cg1_batch_decompress_device produce
 * them; zeros = identity) -> n compressed48 (to_compressed_bytes, util.py:27-28,120).  All device pointers. */
int cg1_batch_compress_device(cg1_ctx* ctx, const void* d_in_affine96, void* d_out48, size_t n);
/* deterministic synthetic scalars, uniform in [1, r-1] (util.py:21-24 distribution), from a 64-bit seed
 * (splitmix64 + rejection), device memory */
int cg1_gen_scalars_device(cg1_ctx* ctx, void* d_out_scalars32, size_t n, uint64_t seed);
/* roofline probe for the dominant kernel: `iters` dependent mixed adds per lane on `lanes` lanes;
 * returns elapsed ms in *ms */
int cg1_probe_madd(cg1_ctx* ctx, const void* d_points_affine96, size_t npts, size_t lanes, int iters, float* ms);

/* ---------------- deferred evaluation of the G1Point operators (north_star: "the prover/verifier keep their Python control flow
 * unchanged") -----------------------------------------------------------------------------------------------------------
 * The reference's loops call the operators one at a time: 585 x from_compressed_bytes_unchecked per verification
 * (whisk_interface.py:96-106 -> util.py:35-36), G_L[i] + G_R[i] * gamma (ipa.py:142-146, same_msm.py:122-126), R * k
 * (curdleproofs.py:310-311), G_i * beta^-i (grand_prod.py:64-71).  The Python face (py_arkworks_bls12381.py) answers them with
 * deferred values and evaluates whole batches through the entry points below when bytes or a comparison are asked for.
 *
 * cg1_validate_compressed: what from_compressed_bytes_unchecked must decide AT THE CALL (util.py:35-36 raises there;
 * test_curdleproofs.py:170-176,211-213): compression flag, x < p, and x^3 + 4 a square -- by its Jacobi symbol, no square root.
 * CG1_OK (is_identity: the infinity flag was set), CG1_ERR_ENCODING, CG1_ERR_NOT_ON_CURVE. */
int cg1_validate_compressed(const uint8_t in48[48], int* is_identity);
int cg1_fp_jacobi(const uint8_t le48[48]);               /* Jacobi symbol (a / p) of a 48-byte little-endian a < p: 1, -1, 0; 2 = a >= p */
/* n encodings (already validated, or not: the first failing index / status come back) -> blobs and / or affine96, on the worker pool */
int cg1_batch_decompress_pool(const uint8_t* in48, size_t n, uint8_t* out_blobs144, uint8_t* out_affine96, int n_threads, size_t* bad_index);
/* the same on the GPU for up to 8 192 encodings (k_batch_decompress_row: one DPP row per point, the square-root chain with one limb per
 * lane; ~0.15 ms whatever n is, through the context's mapped scratch): what the deferred layer uses for batches of >= 192 encodings */
int cg1_batch_decompress_rows(cg1_ctx* ctx, const uint8_t* in48, size_t n, uint8_t* out_blobs144, uint8_t* out_affine96, size_t* bad_index);
/* out_flags[i] = 1 iff affine96 point i (on the curve; zeros = identity) lies in the prime-order subgroup G1: [z^2]P == phi(P) + P.
 * Only for such bases may the coefficient of a deferred product be reduced mod r (`(P * a) * b` == P * (a b mod r)). */
int cg1_batch_subgroup_pool(const uint8_t* affine96, size_t n, uint8_t* out_flags, int n_threads);
/* the same flags; 32 .. 4 096 points and a GPU context: one wave per point with one limb per lane (k_subgroup_row: ~0.25 ms whatever n is);
 * otherwise the pool.  *on_device (may be NULL): 1 when the GPU served the call.  Coordinates >= p: CG1_ERR_ENCODING on both paths. */
int cg1_batch_subgroup(cg1_ctx* ctx, const uint8_t* affine96, size_t n, uint8_t* out_flags, int* on_device);
/* out_j = sum_{t in [offsets[j], offsets[j+1])} scalars32[t] * (+/-) bases[term_base[t] & 0x7fffffff]  (bit 31 of term_base: the negated
 * base); offsets: n_out + 1 host entries, offsets[0] == 0.  path 0 = choose from the batch (never from the machine), 1 = the host's worker
 * pool, 2 = the GPU (cg1_msm_batched_device over the gathered terms).  ctx may be NULL for paths 0 / 1 (then always the pool).  Outputs
 * (each may be NULL) are normalised: point blobs with Z = 1, affine96, compressed48.  *path_used: 1 pool, 2 GPU, 3 = both at once
 * (combinations of >= 4 weighted terms in one k_msm_small launch, the smaller ones on the pool while it runs). */
int cg1_lincomb_batch(cg1_ctx* ctx, const uint8_t* bases_affine96, size_t n_bases, const uint32_t* offsets, size_t n_out,
                      const uint32_t* term_base, const uint8_t* term_scalars32, int path, uint8_t* out_blobs144, uint8_t* out_affine96,
                      uint8_t* out_comp48, int* path_used);
int cg1_lincomb_batch_pool(const uint8_t* bases_affine96, size_t n_bases, const uint32_t* offsets, size_t n_out, const uint32_t* term_base,
                           const uint8_t* term_scalars32, uint8_t* out_blobs144, uint8_t* out_affine96, uint8_t* out_comp48, int n_threads);

/* ---------------- the endomorphism split (csrc/glv.h) -----------------------------------------------------------------------
 * For P in the prime-order subgroup phi(x, y) = (beta x, y) equals lambda P, lambda = z^2 - 1.  With the context parameter "glv" != 0 the
 * CALLER VOUCHES that every point of its MSM calls lies in G1 (CRS points, outputs of earlier MSMs, points that passed the subgroup
 * test); the engine may then run an MSM over the 2n entries (k1, P_i), (k2, phi(P_i)) of the 127-bit halves k = k1 + k2 lambda: the same
 * bucket additions, half the windows (half the doublings of the Horner tail).  1 = where it pays: the single-launch kernel (n <= 1 024,
 * also the <= 64 MSMs of one cg1_msm_batched* / cg1_lincomb_batch launch), the regime-B chain, and regime A up to "glv_max_n" terms (2^14: above, the second
 * half of the table costs more than the tail gains, profiles/r05_glv_ab.txt); 2 = wherever it can (A/B runs).  Outside G1
 * phi(P) != lambda P and the result would be wrong: the default is 0, and the Python face switches it on for single calls whose bases it
 * has certified.
 * cg1_glv_split: the split of one scalar (< 2^255, little-endian) as the digit kernels compute it -- magnitudes < 2^127 and signs --
 * for tests: (-1)^neg1 k1 + (-1)^neg2 k2 lambda == k (mod r). */
void cg1_glv_split(const uint8_t scalar32[32], uint8_t k1_16[16], uint8_t k2_16[16], int* neg1, int* neg2);

/* The lone-wave addition probe (csrc/fp_row.h): `waves` waves each run `iters` DEPENDENT EC additions; mode 0 = one lane per addition (the
 * formulas k_accumulate uses), 1 = one DPP quad per addition (g1_quad.h: the latency-bound kernels), 2 = one limb per lane, the four
 * products of a stage on the four rows of the wave.  *ms = device time of one launch (best of reps); out_blob144 = wave 0's result. */
int cg1_probe_add_chain(cg1_ctx* ctx, int mode, const uint8_t* two_points_affine96, size_t waves, int iters, int reps,
                        uint8_t* out_blob144, float* ms);

/* ---------------- native Merlin transcript (host C++) ------------------------
 * Stands behind merlin_transcripts/merlin_transcripts/{merlin_transcript.py:6-24, strobe.py:16-107,
 * keccak.py:16-66} and curdleproofs/curdleproofs/curdleproofs_transcript.py:7-28.
 * `state` is a caller-owned CG1_MERLIN_STATE_BYTES blob; copying the blob forks the transcript. */
#define CG1_MERLIN_STATE_BYTES 208
void cg1_keccak_f1600(uint8_t* state200);              /* keccak.py:16-66, one permutation */
void cg1_keccak_f1600_x8(uint64_t* lanes);               /* eight states at once, lane w of state k at lanes[8*w + k] (AVX-512 when present) */
void cg1_keccak_f1600_x8_states(uint8_t* const* states200, int live);   /* 1..8 sponges permuted where they lie */
void cg1_strobe_new(uint8_t* state, const uint8_t* protocol_label, size_t len);                 /* Strobe128.new */
int  cg1_strobe_meta_ad(uint8_t* state, const uint8_t* data, size_t len, int more);
int  cg1_strobe_ad(uint8_t* state, const uint8_t* data, size_t len, int more);
int  cg1_strobe_prf(uint8_t* state, uint8_t* out, size_t len, int more);
int  cg1_strobe_key(uint8_t* state, const uint8_t* data, size_t len, int more);
void cg1_merlin_init(uint8_t* state, const uint8_t* label, size_t len);                          /* MerlinTranscript(label) */
void cg1_merlin_append(uint8_t* state, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len);
void cg1_merlin_append_list(uint8_t* state, const uint8_t* label, size_t label_len, const uint8_t* items,
                            size_t item_len, size_t count);                                       /* append_list */
void cg1_merlin_challenge(uint8_t* state, const uint8_t* label, size_t label_len, uint8_t* out, size_t out_len);
/* get_and_append_challenge: 32 LE bytes of a canonical non-zero Fr element, already re-appended */
void cg1_merlin_challenge_scalar(uint8_t* state, const uint8_t* label, size_t label_len, uint8_t out32[32]);

/* ---------------- batch verifier front-end of the shuffle argument (host C++) ----------
 * Stands behind CurdleProofsProof.verify (curdleproofs/curdleproofs/curdleproofs.py:160-246) and its
 * sub-verifiers same_perm.py:75-121, grand_prod.py:161-218, ipa.py:156-236, same_scalar.py:71-111,
 * same_msm.py:146-227, on the wire format of WhiskShuffleProof.from_bytes (whisk_interface.py:64-69) and of the
 * trackers (whisk_interface.py:96-100).  It does NOT verify by itself: it reduces each proof to the statement
 *        sum_k own_scalars[k] * own_points[k]  +  sum_j crs_scalars[j] * crs_points[j]  ==  identity
 * which the caller checks with cg1_batch_decompress_device + cg1_msm_device (many proofs merged: add the CRS
 * scalar vectors) or cg1_msm_batched_device (independently).  Python driver: curdleproofs_pie_amd/shuffle_verifier.py.
 *
 * Layouts (ell trackers, n = ell + 4 = 2^lg):
 *   crs bytes       (ell+9) x 48: vec_G | vec_H(4) | H | G_t | G_u | G_sum | H_sum  = CurdleproofsCrs.to_bytes (crs.py:92-101)
 *   instance        4*ell x 48:   vec_R | vec_S | vec_T | vec_U (tracker r_G / k_r_G encodings, pre then post)
 *   proof           cg1_shuffle_proof_bytes(): M | A | cm_T | cm_U | R | S | same_perm | same_scalar | same_msm
 *   own points      cg1_shuffle_points_per_proof() = 4*ell + 19 + 10*lg encodings: the instance, then the proof's
 *                   points in wire order; own scalars use the same indexing, crs scalars the crs-bytes indexing
 *   weights         12 x scalar32 per proof: the random rho of each of the 8 accumulated checks
 *                   (msm_accumulator.py:43) and of the 4 same-scalar equalities (same_scalar.py:108)
 *   status          0 = prepared; CG1_SHUFFLE_* = rejected before any group arithmetic (its scalars are zeroed)
 *   challenges      optional (may be NULL), for parity tests: alpha_p beta_p alpha_g beta_g alpha_ipa beta_ipa
 *                   alpha_samescalar alpha_samemsm | gamma_ipa[lg] | gamma_samemsm[lg] | vec_a[ell]
 */
#define CG1_SHUFFLE_BAD_SCALAR   1   /* an Fr field of the proof is >= r   (Scalar.from_le_bytes raises, util.py:149-153) */
#define CG1_SHUFFLE_BAD_POINT    2   /* A, cm_T.T_1, cm_U.T_1 or B does not decode (util.py:143-147)                     */
#define CG1_SHUFFLE_T0_INFINITY  3   /* vec_T[0] is the identity (curdleproofs.py:173-174)                               */
#define CG1_SHUFFLE_BAD_WEIGHT   4   /* a caller-supplied weight is >= r                                                  */
typedef struct cg1_shuffle_crs cg1_shuffle_crs;
cg1_shuffle_crs* cg1_shuffle_crs_create(const uint8_t* crs_bytes, size_t ell, size_t n_blinders);   /* NULL: bad sizes / encoding */
void   cg1_shuffle_crs_destroy(cg1_shuffle_crs* crs);
size_t cg1_shuffle_proof_bytes(const cg1_shuffle_crs* crs);
size_t cg1_shuffle_points_per_proof(const cg1_shuffle_crs* crs);
size_t cg1_shuffle_crs_points(const cg1_shuffle_crs* crs);
size_t cg1_shuffle_challenges_per_proof(const cg1_shuffle_crs* crs);
/* decoded96 (may be NULL): per proof, at decoded96 + i*decoded_stride, the 8 consecutive own points 4*ell+1 .. 4*ell+8
 * (A T_1 T_2 U_1 U_2 R S B) in affine96 form as cg1_batch_decompress_device produced them (one strided D2H copy).
 * Of these the verifier needs A, T_1, U_1, B as group elements (A' = A + T_1 + U_1, curdleproofs.py:204; D,
 * grand_prod.py:186).  NULL: the host decodes those four itself (4 square roots per proof). */
int cg1_shuffle_prepare(const cg1_shuffle_crs* crs, size_t n_proofs, const uint8_t* instances, const uint8_t* proofs,
                        const uint8_t* weights, const uint8_t* decoded96, size_t decoded_stride,
                        uint8_t* out_points48, uint8_t* out_scalars32, uint8_t* out_crs_scalars32,
                        int32_t* status, uint8_t* out_challenges32, int n_threads /* 0 = all cores */);
/* 1 (default): cg1_shuffle_prepare advances 16 proofs' transcripts in step, their Keccak-f permutations eight at a time
 * (AVX-512 when present); 0: one transcript at a time.  Same output bytes either way. */
void cg1_shuffle_set_grouped(int on);
/* threads used when n_threads = 0: usable CPUs (affinity mask capped by the cgroup CPU quota; env CURDLE_G1_THREADS overrides) */
size_t cg1_shuffle_default_threads(void);
/* Batched Merlin transcripts on the device (HIP half): n transcripts, one per GPU lane, all starting from
 * init_state208 (what cg1_merlin_init left on the host) and all running the same operation list on their own data rows:
 *   kind 0  append_message(label, data_row[data_off .. data_off + len))                      merlin_transcript.py:11-15
 *   kind 1  challenge_bytes(label, len) -> out_row[out_off ..]                                merlin_transcript.py:20-24
 *   kind 2  get_and_append_challenge(label) -> 32 bytes at out_row[out_off ..]                curdleproofs_transcript.py:15-25
 *   kind 3  append_message(label, out_row[out_off .. out_off + len))   (a value the transcript produced itself)
 * d_states_out (optional): the n final 208-byte state blobs, interchangeable with the host functions above. */
typedef struct cg1_merlin_op {
  uint8_t kind, label_len;
  uint16_t pad;
  uint32_t len, data_off, out_off;
  uint8_t label[32];
} cg1_merlin_op;
/* Keccak passes the slowest wave of the last cg1_merlin_batch_device call executed (the lanes of a wave permute together;
 * a shuffle-shaped program needs ~750 permutations per transcript). */
int cg1_merlin_last_passes(const cg1_ctx* ctx);
/* Which kernel served that call: 2 = block program (whole rate blocks per pass; the default when the operation list fits its row format:
 * challenges of at most 164 bytes, 4-byte aligned output offsets, at most four self-produced pieces per block), 1 = byte-level state
 * machine ("merlin_rows" = 0, or the program does not fit), 0 = one lane at a time ("merlin_sync" = 0, or more than 48 distinct labels). */
int cg1_merlin_last_kernel(const cg1_ctx* ctx);
/* Host only, test support: ONE transcript of that interface run through the block program on the CPU (the tables the device kernels
 * consume, walked the way they walk them); CG1_ERR_ARG when the operation list does not fit the row format. */
int cg1_merlin_block_program_emulate(const uint8_t* init_state208, const cg1_merlin_op* ops, size_t nops, const uint8_t* data_row, size_t data_bytes,
                                     uint8_t* out_row, size_t out_bytes, uint8_t* state_out208 /* may be NULL */, uint32_t* passes);
int cg1_merlin_batch_device(cg1_ctx* ctx, const uint8_t* init_state208, const cg1_merlin_op* ops, size_t nops, const void* d_data,
                            size_t data_stride, void* d_out, size_t out_stride, void* d_states_out, size_t n);

/* Scalar rows on the device (ipa.py:155-186,216,227-229; same_msm.py:146-182,213; grand_prod.py:64-71;
 * msm_accumulator.py:43-58).  cg1_shuffle_prepare_inputs is cg1_shuffle_prepare without the row expansion: per proof it emits
 * cg1_shuffle_rowin_scalars(crs) 32-byte scalars (the transcript's challenges, their inverses, beta^-1, inner_prod, the
 * proof's Fr fields, the weights).  cg1_shuffle_rows_device expands them on the GPU into the same rows, byte for byte:
 * d_out_scalars receives n_proofs x points_per_proof own-point scalars followed by the crs_points sums over the live proofs;
 * d_crs_rows the per-proof CRS rows; d_status_out the final per-proof codes (host code, or CG1_SHUFFLE_BAD_POINT when
 * d_point_status -- the decompression verdicts of the proof's own points -- has a non-zero byte). */
size_t cg1_shuffle_rowin_scalars(const cg1_shuffle_crs* crs);
int cg1_shuffle_prepare_inputs(const cg1_shuffle_crs* crs, size_t n_proofs, const uint8_t* instances, const uint8_t* proofs,
                               const uint8_t* weights, const uint8_t* decoded_affine96, size_t decoded_stride, uint8_t* out_points48,
                               uint8_t* out_rowin32, int32_t* status, int n_threads);
int cg1_shuffle_rows_device(cg1_ctx* ctx, size_t ell, size_t lg, size_t n_proofs, const void* d_rowin, const void* d_host_status,
                            const void* d_point_status, void* d_out_scalars, void* d_crs_rows, void* d_status_out);
/* Exact (unweighted) evaluation on the host of the equalities the reference asserts directly, for proofs that carry a
 * point outside G1 (cg1_subgroup_flags_enqueue / CG1_ERR_NOT_IN_SUBGROUP): the four same-scalar equalities of one shuffle
 * proof (same_scalar.py:101-108; *ok = 1 iff all hold), and the two equalities of one tracker-opening proof
 * (opening.py:73-76).  Points are decoded unchecked, scalars act as integers in [0, r), as in the reference. */
/* The front-end ON THE DEVICE (csrc/kernels_frontend.h): what cg1_shuffle_prepare_inputs does on the host -- the Fiat-Shamir transcript
 * (curdleproofs_transcript.py:15-25 over merlin_transcripts), the grand-product scalar (same_perm.py:98-101), D (grand_prod.py:157) and
 * A' (curdleproofs.py:210) with their encodings, inner_prod (grand_prod.py:164-166), the challenge inverses -- one proof per lane on the
 * wire points already in HBM.  Same outputs, byte for byte: the row-input blocks and the per-proof front-end codes.
 * cg1_shuffle_gather_aux collects what the kernel needs besides the points: per proof r_p c_final d_final z_k z_t z_u x_final | 12 weights. */
typedef struct cg1_shuffle_fe cg1_shuffle_fe;
cg1_shuffle_fe* cg1_shuffle_fe_create(cg1_ctx* ctx, size_t ell, size_t lg, const uint8_t* crs_affine96, const uint8_t* crs48);   /* NULL on failure */
void   cg1_shuffle_fe_destroy(cg1_shuffle_fe* fe);
size_t cg1_shuffle_fe_aux_bytes(void);
/* The kernel runs a BLOCK PROGRAM: the transcript of a given ell cut on the host into the rate blocks between two permutations
 * ("nodes"; 0 = this ell does not fit the format and the byte-level state machine is used; cg1_ctx_set_param("fe_rows", 0) forces
 * that one).  _last_passes: Keccak passes of the slowest wave of the last launch enqueued on ctx (waits for the stream). */
size_t cg1_shuffle_fe_nodes(const cg1_shuffle_fe* fe);
int    cg1_shuffle_fe_emulate_to_first_barrier(size_t ell, size_t lg, const uint8_t* crs_h48, const uint8_t* wire, uint8_t* out_row, size_t out_row_bytes,
                                               uint32_t* passes);   /* host only, test support: the block program of ONE proof walked on the CPU up to the grand-product step; out_row >= (cg1_shuffle_rowin_scalars + 6) * 32 bytes, the drawn challenges in their slots */
int    cg1_shuffle_fe_program_shape(size_t ell, size_t lg, uint32_t* out4);   /* host only: {operations, nodes (0 = does not fit), squeeze nodes, max late pieces per node} */
size_t cg1_shuffle_fe_last_passes(cg1_shuffle_fe* fe, cg1_ctx* ctx);
void   cg1_shuffle_fe_last_split(const cg1_shuffle_fe* fe, uint32_t* out7);   /* ("fe_timed" launches) that wave's shader clocks / 256: late pieces + row loads issued | Keccak-f | whole passes | draw + range check | X_GPROD | X_DA | X_FINAL */
int cg1_shuffle_gather_aux(const cg1_shuffle_crs* crs, size_t n_proofs, const uint8_t* proofs, const uint8_t* weights, uint8_t* out_aux);
int cg1_shuffle_fe_enqueue(cg1_shuffle_fe* fe, cg1_ctx* ctx, size_t n, const void* d_wire48, const void* d_pts_affine96, const void* d_aux,
                           void* d_rowin, void* d_status, int lanes_per_wave /* 1..64 transcripts per wave; 0 = 64 */);
int cg1_shuffle_exact_same_scalar(const cg1_shuffle_crs* crs, const uint8_t* instance, const uint8_t* proof, int* ok);
int cg1_opening_exact(const uint8_t* tracker96 /* r_G | k_r_G */, const uint8_t* k_commitment48, const uint8_t* proof128, int* ok);
/* the same check with the batch path's status code: 0 accepted, CG1_SHUFFLE_BAD_SCALAR, CG1_SHUFFLE_BAD_POINT, 6 = an equality fails.  What
 * OpeningBatchVerifier uses for a handful of proofs -- IsValidWhiskOpeningProof itself (whisk_interface.py:147-169) is a batch of ONE: five
 * single decompressions and four scalar multiplications on the host (~0.4 ms) against ~2.4 ms of dependent GPU launches. */
int cg1_opening_exact_status(const uint8_t* tracker96, const uint8_t* k_commitment48, const uint8_t* proof128, int* status);
/* just the gather step: every proof's own points (instance, then the proof's points in wire order) */
int cg1_shuffle_gather_points(const cg1_shuffle_crs* crs, size_t n_proofs, const uint8_t* instances, const uint8_t* proofs,
                              uint8_t* out_points48);
/* fold cg1_batch_decompress_device's per-point status bytes (n_proofs x points_per_proof) into status[]: a proof
 * with an undecodable point becomes CG1_SHUFFLE_BAD_POINT and its scalars are zeroed (BufReader.read_g1, util.py:143-147) */
int cg1_shuffle_apply_point_status(int32_t* status, const uint8_t* point_status, size_t n_proofs, size_t points_per_proof,
                                   uint8_t* scalars32, uint8_t* crs_scalars32, size_t ncrs);
/* out[j] = sum over proofs i with status[i] == 0 (status may be NULL = all) of crs_scalars[i][j]  (mod r) */
int cg1_shuffle_sum_crs_scalars(const uint8_t* crs_scalars32, const int32_t* status, size_t n_proofs, size_t ncrs, uint8_t* out32);

/* Whisk tracker-opening proofs in batches: IsValidWhiskOpeningProof (whisk_interface.py:147-169) ->
 * TrackerOpeningProof.verify (opening.py:60-79), wire format opening.py:94-106.  Own points per proof, in order:
 * k_G, k_r_G, r_G, A, B (5 encodings / 5 scalars); out_g_scalars32 holds each proof's scalar on the generator G
 * (add them up with cg1_shuffle_sum_crs_scalars(..., ncrs = 1)).  status: 0 or CG1_SHUFFLE_BAD_SCALAR / _BAD_WEIGHT;
 * undecodable points are found by cg1_batch_decompress_device (apply with cg1_shuffle_apply_point_status, 5 points per proof). */
int cg1_opening_prepare(size_t n, const uint8_t* trackers96 /* r_G | k_r_G */, const uint8_t* k_commitments48, const uint8_t* proofs128,
                        const uint8_t* weights /* n x 2 x scalar32 */, uint8_t* out_points48, uint8_t* out_scalars32,
                        uint8_t* out_g_scalars32, int32_t* status);

/* The same front-end on the device (csrc/kernels_opening.h; replaces the per-proof host work of opening.py:60-71 -- the transcript --
 * and of opening.py:73-74 as one random combination): host wire bytes in, d_points96 ((5 n + 1) x 96 B: the own points decoded WITH the
 * subgroup test, then the generator) and d_scalars32 ((5 n + 1) x 32 B, the last one the summed generator scalar) left on the device for
 * cg1_msm_device.  status[n]: 0 or CG1_SHUFFLE_BAD_SCALAR / _BAD_WEIGHT / _BAD_POINT (scalars of such a proof are zero); point_status[5 n]:
 * k_batch_decompress's codes (CG1_ERR_NOT_IN_SUBGROUP marks the proofs the exact check cg1_opening_exact decides);
 * out_g_scalars32 (n x 32 B, may be NULL): each proof's own generator scalar, for the culprit search.  No weight in the reference: it
 * asserts both equalities per proof; the random combination is the batch verifier's own. */
int cg1_opening_prepare_device(cg1_ctx* ctx, size_t n, const uint8_t* trackers96 /* r_G | k_r_G */, const uint8_t* k_commitments48,
                               const uint8_t* proofs128, const uint8_t* weights /* n x 2 x scalar32, or NULL */, const uint8_t* seed32 /* used when weights == NULL */,
                               void* d_points96, void* d_scalars32, int32_t* status, uint8_t* point_status, uint8_t* out_g_scalars32);
/* The weights both front-ends use when the caller supplies none: proof i gets rho1 | rho2 = two 128-bit values (as scalar32) from the first
 * 32 bytes of SHAKE256(seed32 || le64(i)); seed32 = 32 fresh bytes from the OS per batch.  Writes proofs first .. first + n - 1. */
int cg1_opening_weights_from_seed(const uint8_t seed32[32], size_t first, size_t n, uint8_t* out_weights64);

#ifdef __cplusplus
}
#endif
#endif /* CURDLE_G1_H */
