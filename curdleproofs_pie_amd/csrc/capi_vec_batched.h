// C ABI, part 2: resident point vectors, the batched normalisation, regime B.
// Part of the single translation unit csrc/msm_gpu.hip (included there, in this order; not a stand-alone header).
#pragma once

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
}  // extern "C"
