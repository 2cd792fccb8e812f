// k_accumulate (the dominant kernel) and the chunk-sum joiners k_heavy_combine / k_bucket_fold.
// Part of the single translation unit csrc/msm_gpu.hip (included inside namespace cg1).
#pragma once

// ------------------------------------------------------------------ k_accumulate (dominant kernel)
__global__ void __launch_bounds__(256) k_accumulate(const uint2* __restrict__ desc, const uint32_t* __restrict__ total_chunks,
                                                    const uint32_t* __restrict__ order, const uint32_t* __restrict__ sorted,
                                                    const PreparedPoint* __restrict__ pts, PointSum* __restrict__ sums) {
  uint32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g >= *total_chunks) return;
  const uint32_t t = order[g];              // chunks in descending length: lanes of a wave finish together
  const uint2 d = desc[t];
  const uint32_t* ent = sorted + d.x;
  if (d.y == 0u) { store_sum(sums + t, xyzz_identity()); return; }     // k_chunk_desc emits no empty chunks; kept for safety
  uint32_t e = ent[0];
  fp x, y; uint32_t flags;
  load_affine(pts + (e & 0x7fffffffu), x, y, flags);
  xyzz acc;
  uint32_t j = 1;
  if (e >> 31) y = fp_neg<3>(y);
  if (d.y >= 2u) {
    // the first addition of a chunk has two affine operands: 4M+2S instead of 8M+2S
    const uint32_t e1 = ent[1];
    fp x1, y1;
    load_affine(pts + (e1 & 0x7fffffffu), x1, y1, flags);
    if (e1 >> 31) y1 = fp_neg<3>(y1);
    e = ent[d.y > 2u ? 2 : 1];
    fp xn, yn;
    load_affine(pts + (e & 0x7fffffffu), xn, yn, flags);          // prefetch entry 2 while the pair is being added
    acc = xyzz_mmadd(x, y, x1, y1);
    x = xn; y = yn;
    j = 2;
  } else {
    acc = xyzz_from_affine(x, y);
  }
  for (; j < d.y; ++j) {
    // prefetch the next entry's point while this one is being added
    const uint32_t en = ent[(j + 1 < d.y) ? j + 1 : j];
    fp xn, yn;
    load_affine(pts + (en & 0x7fffffffu), xn, yn, flags);
    if (e >> 31) y = fp_neg<3>(y);
    if (!xyzz_madd_fast(acc, x, y)) break;                  // doubling / cancellation / identity accumulator: below
    e = en; x = xn; y = yn;
  }
  for (; j < d.y; ++j) {                                    // the rest of a chunk that met an exceptional case (y already signed)
    acc = xyzz_madd(acc, x, y);
    if (j + 1 < d.y) {
      e = ent[j + 1];
      load_affine(pts + (e & 0x7fffffffu), x, y, flags);
      if (e >> 31) y = fp_neg<3>(y);
    }
  }
  store_sum(sums + t, acc);
}

// ------------------------------------------------------------------ k_heavy_combine
// Blocks stride over the heavy-bucket list; one block adds ALL chunk sums of its bucket (<= 4096):
// <= 16 serial adds per lane, then wave shuffles, then LDS.  The total replaces the bucket's first chunk sum and
// the bucket is flagged so k_seg_reduce reads only that slot.
__global__ void __launch_bounds__(256) k_heavy_combine(const uint32_t* __restrict__ heavy, uint32_t heavy_cap,
                                                       const uint32_t* __restrict__ choff, PointSum* __restrict__ sums,
                                                       uint8_t* __restrict__ combined) {
  __shared__ PointSum sh[4];
  uint32_t nheavy = heavy[0];
  if (nheavy > heavy_cap) nheavy = heavy_cap;
  for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
    const uint32_t b = heavy[1 + h];
    const uint32_t c0 = choff[b], c1 = choff[b + 1];
    xyzz acc = xyzz_identity();
    for (uint32_t k = c0 + threadIdx.x; k < c1; k += 256) acc = xyzz_add(acc, load_sum(sums + k));
    __syncthreads();                     // every chunk sum has been read before slot c0 is overwritten
    for (int delta = 32; delta >= 1; delta >>= 1) {
      xyzz o = shfl_down_xyzz(acc, delta);
      if ((threadIdx.x & 63) < (uint32_t)delta) acc = xyzz_add(acc, o);
    }
    if ((threadIdx.x & 63) == 0) store_sum(&sh[threadIdx.x >> 6], acc);
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int k = 1; k < 4; ++k) acc = xyzz_add(acc, load_sum(&sh[k]));
      store_sum(sums + c0, acc);
      combined[b] = 1;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ k_bucket_fold
// One lane per bucket: buckets whose entries were cut into 2..16 chunks (a window-sharded rank owns few buckets,
// so chunks are kept short for parallelism in k_accumulate) get their chunk sums added serially into the first
// slot.  Buckets with more chunks were already handled by k_heavy_combine; single-chunk buckets are untouched.
__global__ void __launch_bounds__(256) k_bucket_fold(const uint32_t* __restrict__ choff, PointSum* __restrict__ sums,
                                                     uint8_t* __restrict__ combined, uint32_t nb_total, const uint32_t* __restrict__ any_multi = nullptr) {
  if (any_multi && *any_multi == 0u) return;       // no bucket was cut into 2..16 chunks (uniform scalars on one GPU): nothing to fold
  uint32_t b = blockIdx.x * 256 + threadIdx.x;
  if (b >= nb_total) return;
  const uint32_t c0 = choff[b], c1 = choff[b + 1];
  if (c1 - c0 < 2u || c1 - c0 >= HEAVY_MIN_CHUNKS) return;
  xyzz acc = load_sum(sums + c0);
  for (uint32_t k = c0 + 1; k < c1; ++k) acc = xyzz_add(acc, load_sum(sums + k));
  store_sum(sums + c0, acc);
  combined[b] = 1;
}

