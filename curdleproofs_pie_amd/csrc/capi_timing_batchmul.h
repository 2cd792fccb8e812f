// C ABI, part 4: timings and counters of the last call, the batched scalar multiplication.
// Part of the single translation unit csrc/msm_gpu.hip (included there, in this order; not a stand-alone header).
#pragma once

extern "C" {
int cg1_get_timings(const cg1_ctx* ctx, float* phase_ms, float* host_tail_ms, int* window_c) {
  if (!ctx) return CG1_ERR_ARG;
  if (phase_ms) for (int i = 0; i < CG1_NPHASE; ++i) phase_ms[i] = ctx->phase_ms[i];
  if (host_tail_ms) *host_tail_ms = ctx->host_tail_ms;
  if (window_c) *window_c = ctx->last_c;
  return CG1_OK;
}

int cg1_get_last_launches(const cg1_ctx* ctx) { return ctx ? ctx->last_acc_launches : -1; }

// The window plan the engine would use (no GPU needed): (offset, width) of every window; returns the window count, or -1.
int cg1_plan_describe(int window_c, int glv, int* out_offsets, int* out_widths, int capacity) {
  const int cabs = window_c < 0 ? -window_c : window_c;
  if (cabs < 4 || cabs > 16) return -1;
  const cg1::WinPlan pl = cg1::make_plan(window_c, glv != 0);
  if (pl.nwin > capacity) return -1;
  for (int w = 0; w < pl.nwin; ++w) { out_offsets[w] = pl.off(w); out_widths[w] = pl.width(w); }
  return pl.nwin;
}

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
