// C ABI (include/curdle_g1.h), part 1: single-element host operators, the device context, memory, parameters, the MSM entry points.
// Part of the single translation unit csrc/msm_gpu.hip (included there, in this order; not a stand-alone header).
#pragma once

// ================================================================== C ABI (include/curdle_g1.h)
using cg1::Ctx;
struct cg1_ctx : public cg1::Ctx {};

namespace cg1 {
static Ctx* child_of(Ctx* ctx) { return static_cast<Ctx*>(ctx->child); }

// One call as two launch chains: this context takes the HIGH half of the plan's windows (and prepares the points), its child the
// LOW half on its own stream.  The child starts once the prepared records exist, and its k_accumulate waits for this context's to
// finish: the two dominant launches run back to back, everything around them overlaps with one of them.
static int ensure_child(Ctx* ctx) {
  if (!ctx->child) {
    cg1_ctx* made = ctx->cu_mask.empty() ? cg1_ctx_create(ctx->device) : cg1_ctx_create_cu_mask(ctx->device, ctx->cu_mask.data(), ctx->cu_mask.size());
    if (!made) { snprintf(ctx->err, sizeof ctx->err, "could not create the second launch chain's context"); return CG1_ERR_HIP; }
    made->split = 0;
    made->batched_split = 0;
    ctx->child = made;
  }
  if (!ctx->ev_prep) HIPCHK(hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming));
  if (!ctx->ev_acc) HIPCHK(hipEventCreateWithFlags(&ctx->ev_acc, hipEventDisableTiming));
  return CG1_OK;
}

static int msm_begin_split(Ctx* ctx, const PtSrc& src, const void* d_scalars32, size_t n, const WinPlan& plan, int rank, int world) {
  HIPCHK(hipSetDevice(ctx->device));
  { int crc = ensure_child(ctx); if (crc) return crc; }
  Ctx* ch = child_of(ctx);
  ch->profile = ctx->profile; ch->L0 = ctx->L0; ch->seg_m = ctx->seg_m; ch->quad = ctx->quad; ch->reduce_2d = ctx->reduce_2d;
  ch->rowcol_quad = ctx->rowcol_quad; ch->rowcol_quad_max = ctx->rowcol_quad_max; ch->fold_pass = ctx->fold_pass; ch->tree_half = ctx->tree_half; ch->tree_shift = ctx->tree_shift; ch->rowcol_lgq = ctx->rowcol_lgq; ch->tree_row = ctx->tree_row; ch->rowcol_row = ctx->rowcol_row; ch->sort_sub_bits = ctx->sort_sub_bits;
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
  if (!strcmp(name, "glv")) { if (value < 0 || value > 2) return CG1_ERR_ARG; ctx->glv = value; return CG1_OK; }
  if (!strcmp(name, "glv_max_n")) { if (value < 0) return CG1_ERR_ARG; ctx->glv_max_n = value; return CG1_OK; }
  if (!strcmp(name, "batched_split")) { ctx->batched_split = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "batched_split_min_m")) { if (value < 2) return CG1_ERR_ARG; ctx->batched_split_min_m = value; return CG1_OK; }
  if (!strcmp(name, "lincomb_zero_copy")) { ctx->lincomb_zero_copy = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "fold_quad")) { ctx->fold_quad = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "rowcol_row")) { ctx->rowcol_row = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "tree_row")) { ctx->tree_row = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "horner_row")) { ctx->horner_row = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "batch_mul_row")) { ctx->batch_mul_row = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "small_msm")) { ctx->small_msm = value ? 1 : 0; return CG1_OK; }
  if (!strcmp(name, "small_row_tail")) { ctx->small_row_tail = value ? 1 : 0; return CG1_OK; }
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

// the context's mapped page-locked scratch (cg1_lincomb_batch's gather buffer, cg1_msm's small inputs): at least `bytes`
static int ensure_lin(cg1_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->cap_h_lin) return CG1_OK;
  if (ctx->h_lin) (void)hipHostFree(ctx->h_lin);
  ctx->h_lin = nullptr; ctx->h_lin_dev = nullptr; ctx->cap_h_lin = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  HIPCHK(hipHostMalloc((void**)&ctx->h_lin, want, hipHostMallocMapped));
  HIPCHK(hipHostGetDevicePointer((void**)&ctx->h_lin_dev, ctx->h_lin, 0));
  ctx->cap_h_lin = want;
  return CG1_OK;
}

int cg1_msm(cg1_ctx* ctx, const uint8_t* points, const uint8_t* scalars, size_t n, uint8_t* out) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) { blob_out(out, cg1h::jac_identity()); return CG1_OK; }
  HIPCHK(hipSetDevice(ctx->device));
  if (n <= 4096 && ctx->lincomb_zero_copy) {
    // the protocol's own sizes (the accumulator's final 5 ell + 7 terms): two copies from pageable memory cost more than the kernels'
    // reading ~100 KB from mapped host memory
    { int lrc = ensure_lin(ctx, n * 128); if (lrc) return lrc; }
    memcpy(ctx->h_lin, points, n * 96);
    memcpy(ctx->h_lin + n * 96, scalars, n * 32);
    return cg1_msm_device(ctx, ctx->h_lin_dev, ctx->h_lin_dev + n * 96, n, 0, 0, 1, out);
  }
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
