// C ABI, part 3: the batched evaluation of deferred G1Point operators (linear combinations over shared bases, subgroup flags).
// Part of the single translation unit csrc/msm_gpu.hip (included there, in this order; not a stand-alone header).
#pragma once

extern "C" {
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
constexpr size_t LINCOMB_ZERO_COPY_MAX = 4096;   // terms of a GPU share the kernels read from mapped host memory instead of a staged copy
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
// out_flags[i] = 1 iff affine96 point i lies in G1.  32 .. 4 096 points with a GPU context: one wave per point (k_subgroup_row, fp_row.h);
// fewer, more, or no context: the host's worker pool (cg1_batch_subgroup_pool).  *on_device (may be NULL): which one ran.
int cg1_batch_subgroup(cg1_ctx* ctx, const uint8_t* affine96, size_t n, uint8_t* out_flags, int* on_device) {
  if (on_device) *on_device = 0;
  if (n == 0) return CG1_OK;
  if (!affine96 || !out_flags) return CG1_ERR_ARG;
  if (!ctx || n < 32 || n > LINCOMB_ROW_MAX) return cg1_batch_subgroup_pool(affine96, n, out_flags, 0);
  for (size_t i = 0; i < n; ++i) {                         // the device kernel takes canonical coordinates (the pool path refuses others too)
    for (int c = 0; c < 2; ++c) {
      uint64_t w[6];
      memcpy(w, affine96 + 96 * i + 48 * c, 48);
      bool lt = false;
      for (int k = 5; k >= 0; --k) { if (w[k] != cg1::H_P[k]) { lt = w[k] < cg1::H_P[k]; break; } }
      if (!lt) return CG1_ERR_ENCODING;
    }
  }
  HIPCHK(hipSetDevice(ctx->device));
  DevBuf dp, df;
  HIPCHK(dp.alloc(n * 96)); HIPCHK(df.alloc(n + 16));
  HIPCHK(hipMemcpyAsync(dp.p, affine96, n * 96, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(cg1::k_subgroup_row, dim3((unsigned)n), dim3(64), 0, ctx->stream, (const uint32_t*)dp.p, (uint32_t)n, (uint8_t*)df.p);
  HIPCHK(hipMemcpyAsync(out_flags, df.p, n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipGetLastError());
  if (on_device) *on_device = 1;
  return CG1_OK;
}

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
    { int lrc = ensure_lin(ctx, TG * 128 + (G + 1) * 4 + 64); if (lrc) return lrc; }
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
    int rc;
    if (TG <= LINCOMB_ZERO_COPY_MAX && ctx->lincomb_zero_copy) {
      // a handful of small combinations (a halving round's 6 MSMs: ~100 KB): the single-launch kernel reads the gathered terms and the
      // offsets straight from the mapped staging buffer -- three copies (~8 us each, one after the other) less in front of a 0.2 ms launch
      uint32_t* ho = reinterpret_cast<uint32_t*>(ctx->h_lin + ((TG * 128 + 63) & ~(size_t)63));
      memcpy(ho, goffs.data(), (G + 1) * 4);
      const uint8_t* dv = ctx->h_lin_dev;
      rc = cg1::msm_batched_device(ctx, dv, dv + TG * 96, goffs.data(), G, 0, gres, &pending,
                                   reinterpret_cast<const uint32_t*>(dv + ((TG * 128 + 63) & ~(size_t)63)));
    } else {
      { int src = ensure_stage(ctx, TG * 96, TG * 32); if (src) return src; }
      HIPCHK(hipMemcpyAsync(ctx->d_stage_pts, hp, TG * 96, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipMemcpyAsync(ctx->d_stage_sc, hs, TG * 32, hipMemcpyHostToDevice, ctx->stream));
      rc = cg1::msm_batched_device(ctx, ctx->d_stage_pts, ctx->d_stage_sc, goffs.data(), G, 0, gres, &pending);
    }
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
}  // extern "C"
