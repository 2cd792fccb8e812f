// C ABI, part 5: batched wire codec (decompression, subgroup flags), batched Merlin transcripts, the opening proofs' device front-end.
// Part of the single translation unit csrc/msm_gpu.hip (included there, in this order; not a stand-alone header).
#pragma once

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
