// C ABI, part 7: scalar rows, batched compression, synthetic scalars, the roofline probes.
// Part of the single translation unit csrc/msm_gpu.hip (included there, in this order; not a stand-alone header).
#pragma once

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

// n validated (or not) encodings -> blobs and / or affine96 through k_batch_decompress_row, everything in the context's mapped scratch:
// no allocation, no staged copy.  The GPU twin of cg1_batch_decompress_pool (same outputs, same error reporting); n <= 8 192.
int cg1_batch_decompress_rows(cg1_ctx* ctx, const uint8_t* in48, size_t n, uint8_t* out_blobs144, uint8_t* out_affine96, size_t* bad_index) {
  if (!ctx) return CG1_ERR_HIP;
  if (n == 0) return CG1_OK;
  if (n > 8192 || !in48) return CG1_ERR_ARG;
  HIPCHK(hipSetDevice(ctx->device));
  const size_t o_out = (n * 48 + 63) & ~(size_t)63, o_st = o_out + n * 96;
  { int lrc = ensure_lin(ctx, o_st + n + 64); if (lrc) return lrc; }
  memcpy(ctx->h_lin, in48, n * 48);
  hipLaunchKernelGGL(cg1::k_batch_decompress_row, dim3((unsigned)((n + 3) / 4)), dim3(64), 0, ctx->stream, (const uint8_t*)ctx->h_lin_dev,
                     reinterpret_cast<uint32_t*>(ctx->h_lin_dev + o_out), ctx->h_lin_dev + o_st, (uint32_t)n);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  const uint8_t* st = ctx->h_lin + o_st;
  for (size_t i = 0; i < n; ++i) if (st[i]) { if (bad_index) *bad_index = i; return st[i]; }
  if (out_affine96) memcpy(out_affine96, ctx->h_lin + o_out, n * 96);
  if (out_blobs144) return cg1_batch_from_affine96(out_blobs144, ctx->h_lin + o_out, n);
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
