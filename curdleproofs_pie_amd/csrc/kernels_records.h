// Device records (prepared points, XYZZ partial sums, exported words) and their vector load/store helpers.
// Part of the single translation unit csrc/msm_gpu.hip (included inside namespace cg1).
#pragma once

// ------------------------------------------------------------------ device records
struct alignas(16) PreparedPoint {      // 128 B
  uint32_t x[NL];
  uint32_t y[NL];
  uint32_t flags;                       // bit0: identity
  uint32_t pad[3];
};
static_assert(sizeof(PreparedPoint) == 128, "one cache line per point");

struct alignas(16) PointSum {           // 256 B: an XYZZ partial sum
  uint32_t c[4][NL];
  uint32_t inf;
  uint32_t pad[7];
};
static_assert(sizeof(PointSum) == 256, "");

struct alignas(16) PointWords {         // 208 B: canonical standard-form XYZZ (see xyzz_words)
  uint32_t w[4][12];
  uint32_t inf;
  uint32_t pad[3];
};
static_assert(sizeof(PointWords) == 208, "");

__device__ __forceinline__ void load_affine(const PreparedPoint* p, fp& x, fp& y, uint32_t& flags) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = q[i];
  const uint32_t* w = reinterpret_cast<const uint32_t*>(v);
#pragma unroll
  for (int i = 0; i < NL; ++i) { x.l[i] = w[i]; y.l[i] = w[NL + i]; }
  flags = w[2 * NL];
}

__device__ __forceinline__ void store_sum(PointSum* dst, const xyzz& a) {
  uint32_t w[64];
#pragma unroll
  for (int i = 0; i < NL; ++i) { w[i] = a.X.l[i]; w[NL + i] = a.Y.l[i]; w[2 * NL + i] = a.ZZ.l[i]; w[3 * NL + i] = a.ZZZ.l[i]; }
  w[4 * NL] = a.inf;
#pragma unroll
  for (int i = 4 * NL + 1; i < 64; ++i) w[i] = 0;
  uint4* q = reinterpret_cast<uint4*>(dst);
#pragma unroll
  for (int i = 0; i < 16; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

__device__ __forceinline__ xyzz load_sum(const PointSum* src) {
  const uint4* q = reinterpret_cast<const uint4*>(src);
  uint32_t w[60];
#pragma unroll
  for (int i = 0; i < 15; ++i) { uint4 v = q[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
  xyzz a;
#pragma unroll
  for (int i = 0; i < NL; ++i) { a.X.l[i] = w[i]; a.Y.l[i] = w[NL + i]; a.ZZ.l[i] = w[2 * NL + i]; a.ZZZ.l[i] = w[3 * NL + i]; }
  a.inf = w[4 * NL];
  return a;
}


__device__ __forceinline__ xyzz shfl_xyzz(const xyzz& a, int src_lane) {        // a as held by lane src_lane of the wave
  xyzz r;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    r.X.l[i] = __shfl(a.X.l[i], src_lane, 64);
    r.Y.l[i] = __shfl(a.Y.l[i], src_lane, 64);
    r.ZZ.l[i] = __shfl(a.ZZ.l[i], src_lane, 64);
    r.ZZZ.l[i] = __shfl(a.ZZZ.l[i], src_lane, 64);
  }
  r.inf = __shfl(a.inf, src_lane, 64);
  return r;
}

__device__ __forceinline__ xyzz shfl_down_xyzz(const xyzz& a, int delta) {
  xyzz r;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    r.X.l[i] = __shfl_down(a.X.l[i], delta, 64);
    r.Y.l[i] = __shfl_down(a.Y.l[i], delta, 64);
    r.ZZ.l[i] = __shfl_down(a.ZZ.l[i], delta, 64);
    r.ZZZ.l[i] = __shfl_down(a.ZZZ.l[i], delta, 64);
  }
  r.inf = __shfl_down(a.inf, delta, 64);
  return r;
}
