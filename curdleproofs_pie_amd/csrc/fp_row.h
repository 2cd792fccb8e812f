// "One limb per lane" field and group arithmetic for LONE waves (device only; part of the translation unit csrc/msm_gpu.hip).
//
// A wave that is alone on its SIMD issues one dependent VALU instruction per ~5-8 clocks whatever it computes, so what a chain of
// dependent EC additions costs there is the NUMBER of instructions on its critical path.  One lane per field product (fp28.h) puts
// ~470 instructions on that path per product; a DPP quad (g1_quad.h) runs four independent products of an addition side by side but
// still pays ~470 per stage.  Here the 14 limbs of a field element live in 14 lanes of a 16-lane DPP row (one VGPR per element), and
// a Montgomery product is 14 steps of ~10 instructions:
//
//     t   += a * bcast(b, i)                       v_mov_dpp row_newbcast:i + v_mad_u64_u32
//     m    = bcast((t0 * -p^-1) mod 2^28, lane 0)  v_mul_lo_u32 + v_and + v_mov_dpp row_newbcast:0
//     t   += p * m                                  v_mad_u64_u32           (lane 0's low 28 bits are now zero)
//     t    = shl_lane(t mod 2^28) + (t >> 28)       v_and + v_alignbit + v_mov_dpp row_shl:1 + v_add      (division by 2^28, carry-save)
//
// i.e. coarsely integrated operand scanning with the division by 2^28 done as a lane shift.  The value is the one fp_mul computes (the
// same a, b, the same unique multiple of p); the limbs come out "nearly normal" (<= 2^28 + 3), which every consumer below tolerates
// (the padded multiples of p used for lazy subtraction dominate 2^28 + 3 in every limb: static_asserts).  The four rows of a wave run
// the four independent products of each stage of an XYZZ addition (the staging of g1_quad.h), results cross rows with ds_bpermute.
// One wave = one EC addition at a time: this only pays where fewer additions are in flight than a chip full of quads would take -- the
// tails of the reductions (measured: tools/gpu_rowlane_ab.py -> profiles/r05_rowlane_ab.txt).
// Included inside namespace cg1, after kernels_records.h (the probe kernel at the end uses its records).
#pragma once

// ---- lane movement
template <int N>
__device__ __forceinline__ uint32_t row_bcast(uint32_t v) {          // v of lane N of my row
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x150 + N, 0xf, 0xf, true);                 // row_newbcast:N (every source lane exists: no "old" value to keep)
}
__device__ __forceinline__ uint32_t row_down1(uint32_t v) {          // lane j <- lane j + 1 of my row; lane 15 <- 0
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xf, 0xf, true);               // row_shl:1, bound_ctrl:0
}
template <int R>
__device__ __forceinline__ uint32_t from_row(uint32_t v, uint32_t lane16) {      // v as row R holds it, in the same lane of my row
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((R * 16 + lane16) << 2), (int)v);
}

// per-lane constants of a row: limb `lane16` of p and of the padded multiples 3p, 6p, 12p (zero in lanes 14, 15)
struct RowK {
  uint32_t lane16, row, p, kp3, kp6, kp12, one;
};
__device__ __forceinline__ RowK row_constants() {
  RowK k;
  const uint32_t lane = (uint32_t)__lane_id();
  k.lane16 = lane & 15u;
  k.row = (lane >> 4) & 3u;
  uint32_t p = 0, k3 = 0, k6 = 0, k12 = 0, one = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const bool me = k.lane16 == (uint32_t)i;
    p = me ? c_p(i) : p;
    k3 = me ? kp_tab<3>::get(i) : k3;
    k6 = me ? kp_tab<6>::get(i) : k6;
    k12 = me ? kp_tab<12>::get(i) : k12;
    one = me ? fp_one().l[i] : one;
  }
  k.p = p; k.kp3 = k3; k.kp6 = k6; k.kp12 = k12; k.one = one;
  return k;
}

// every limb (below the top one) of the padded multiples dominates a nearly-normal limb
constexpr bool kp_dominates(const uint32_t* t) { for (int i = 0; i < NL - 1; ++i) if (t[i] < (1u << 28) + 8u) return false; return true; }
static_assert(kp_dominates(D_KP3) && kp_dominates(D_KP6) && kp_dominates(D_KP12), "lazy subtraction needs KP limbs >= 2^28 + 8");

// ---- conversions
__device__ __forceinline__ uint32_t row_from_fp(const fp& f, uint32_t lane16) {        // every lane holds the same f; keep my limb
  uint32_t v = 0;
#pragma unroll
  for (int i = 0; i < NL; ++i) v = lane16 == (uint32_t)i ? f.l[i] : v;
  return v;
}
#define CG1_ROW_BC(i) r.l[i] = row_bcast<i>(v);
__device__ __forceinline__ fp row_to_fp(uint32_t v) {                                  // every lane of the row gets the whole element
  fp r;
  CG1_ROW_BC(0) CG1_ROW_BC(1) CG1_ROW_BC(2) CG1_ROW_BC(3) CG1_ROW_BC(4) CG1_ROW_BC(5) CG1_ROW_BC(6)
  CG1_ROW_BC(7) CG1_ROW_BC(8) CG1_ROW_BC(9) CG1_ROW_BC(10) CG1_ROW_BC(11) CG1_ROW_BC(12) CG1_ROW_BC(13)
  return r;
}
#undef CG1_ROW_BC

// ---- field operations on rows.  "Nearly normal": limbs 0..12 <= 2^28 + 3.
// one carry pass: value unchanged, limbs -> (limb mod 2^28) + (carry of the limb below): the carry of lane j goes UP to lane j + 1
// (row_shr:1: lane j <- lane j - 1, lane 0 <- 0); limb 13 is the top limb and keeps everything above 2^364
__device__ __forceinline__ uint32_t row_norm_pass(uint32_t t, uint32_t lane16) {
  const uint32_t keep = lane16 == 13u ? t : (t & LMASK);
  const uint32_t c = lane16 == 13u ? 0u : (t >> 28);
  const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x111, 0xf, 0xf, true);
  return keep + up;
}

#define CG1_ROW_STEP(i)                                                              \
  {                                                                                  \
    const uint32_t bi = row_bcast<i>(b);                                             \
    uint64_t acc = (uint64_t)a * bi + (uint64_t)t;                                   \
    uint32_t m = ((uint32_t)acc * D_PINV) & LMASK;                                   \
    m = row_bcast<0>(m);                                                             \
    acc = (uint64_t)m * k.p + acc;                                                   \
    t = row_down1((uint32_t)acc & LMASK) + (uint32_t)(acc >> 28);                    \
  }
#define CG1_ROW_STEP2(i)                                                             \
  {                                                                                  \
    const uint32_t bi = row_bcast<i>(b), di = row_bcast<i>(d);                       \
    uint64_t acc = (uint64_t)a * bi + (uint64_t)t;                                   \
    acc = (uint64_t)c * di + acc;                                                    \
    uint32_t m = ((uint32_t)acc * D_PINV) & LMASK;                                   \
    m = row_bcast<0>(m);                                                             \
    acc = (uint64_t)m * k.p + acc;                                                   \
    t = row_down1((uint32_t)acc & LMASK) + (uint32_t)(acc >> 28);                    \
  }

// a * b * 2^-392 mod-ish p: the value fp_mul(a, b) returns, nearly normal.  Requires max_limb(a) * max_limb(b) < 2^59.
__device__ __forceinline__ uint32_t row_mul(uint32_t a, uint32_t b, const RowK& k) {
  uint32_t t = 0;
  CG1_ROW_STEP(0) CG1_ROW_STEP(1) CG1_ROW_STEP(2) CG1_ROW_STEP(3) CG1_ROW_STEP(4) CG1_ROW_STEP(5) CG1_ROW_STEP(6)
  CG1_ROW_STEP(7) CG1_ROW_STEP(8) CG1_ROW_STEP(9) CG1_ROW_STEP(10) CG1_ROW_STEP(11) CG1_ROW_STEP(12) CG1_ROW_STEP(13)
  return row_norm_pass(row_norm_pass(t, k.lane16), k.lane16);
}
// (a * b + c * d) * 2^-392: one reduction for both products (fp_mul2).  Requires max(a) max(b) + max(c) max(d) < 2^59.
__device__ __forceinline__ uint32_t row_mul2(uint32_t a, uint32_t b, uint32_t c, uint32_t d, const RowK& k) {
  uint32_t t = 0;
  CG1_ROW_STEP2(0) CG1_ROW_STEP2(1) CG1_ROW_STEP2(2) CG1_ROW_STEP2(3) CG1_ROW_STEP2(4) CG1_ROW_STEP2(5) CG1_ROW_STEP2(6)
  CG1_ROW_STEP2(7) CG1_ROW_STEP2(8) CG1_ROW_STEP2(9) CG1_ROW_STEP2(10) CG1_ROW_STEP2(11) CG1_ROW_STEP2(12) CG1_ROW_STEP2(13)
  return row_norm_pass(row_norm_pass(t, k.lane16), k.lane16);
}
#undef CG1_ROW_STEP
#undef CG1_ROW_STEP2

__device__ __forceinline__ uint32_t row_sel(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint32_t row) {
  const uint32_t lo = (row & 1u) ? v1 : v0, hi = (row & 1u) ? v3 : v2;
  return (row & 2u) ? hi : lo;
}

// value == 0 (mod p) for a lazily reduced row value < kmax * p: the fast path of fp_is_zero_mod_p on limb 0, the rare slow path on the
// one-lane form
__device__ __forceinline__ bool row_is_zero_mod_p(uint32_t v, uint32_t kmax) {
  const uint32_t kk = row_bcast<0>((0u - v * D_PINV) & LMASK);
  if (kk >= kmax) return false;
  return fp_is_zero_mod_p(row_to_fp(v), kmax);
}

// ---- the group: XYZZ coordinates, one VGPR each; all four rows of the wave hold the same point
struct xyzz_row { uint32_t X, Y, ZZ, ZZZ, inf; };

__device__ __forceinline__ xyzz_row row_from_xyzz(const xyzz& a, uint32_t lane16) {
  xyzz_row r;
  r.X = row_from_fp(a.X, lane16); r.Y = row_from_fp(a.Y, lane16); r.ZZ = row_from_fp(a.ZZ, lane16); r.ZZZ = row_from_fp(a.ZZZ, lane16);
  r.inf = a.inf;
  return r;
}
__device__ __forceinline__ xyzz row_to_xyzz(const xyzz_row& a, uint32_t lane16) {
  xyzz r;
  // a stored xyzz has strictly normal limbs (g1_xyzz.h): finish the carries of the nearly-normal row values on the one-lane form
  r.X = fp_norm(row_to_fp(a.X)); r.Y = fp_norm(row_to_fp(a.Y)); r.ZZ = fp_norm(row_to_fp(a.ZZ)); r.ZZZ = fp_norm(row_to_fp(a.ZZZ));
  r.inf = a.inf;
  (void)lane16;
  return r;
}

// the exceptional additions (P + P, P - P): decided and computed by the one-lane formulas.  Out of line on purpose: it is ~40 KB of code
// that a kernel with a row_add in its hot path should not carry at every call site (arguments and result travel in registers).
__device__ __attribute__((noinline)) xyzz_row row_add_exceptional(xyzz_row a, xyzz_row b, uint32_t lane16) {
  return row_from_xyzz(xyzz_add(row_to_xyzz(a, lane16), row_to_xyzz(b, lane16)), lane16);
}

// a + b (both XYZZ, any points): the stages of quad_add, one product per row
__device__ __forceinline__ xyzz_row row_add(const xyzz_row& a, const xyzz_row& b, const RowK& k) {
  if (a.inf) return b;
  if (b.inf) return a;
  const uint32_t l = k.lane16, q = k.row;
  // stage 1: row 0: U1 = X1 ZZ2;  row 1: U2 = X2 ZZ1;  row 2: S1 = Y1 ZZZ2;  row 3: S2 = Y2 ZZZ1
  const uint32_t m1 = row_mul(row_sel(a.X, b.X, a.Y, b.Y, q), row_sel(b.ZZ, a.ZZ, b.ZZZ, a.ZZZ, q), k);
  const uint32_t U1 = from_row<0>(m1, l), U2 = from_row<1>(m1, l), S1 = from_row<2>(m1, l), S2 = from_row<3>(m1, l);
  const uint32_t P = U2 + (k.kp3 - U1), R = S2 + (k.kp3 - S1);            // fp_sub<3>: limbs < 2^28 + 2^29, value < 4.1p
  if (row_is_zero_mod_p(P, 6)) {                                           // P + P or P - P: the one-lane formulas decide (rare)
    return row_add_exceptional(a, b, l);
  }
  // stage 2: PP = P P;  RR = R R;  ZZ12 = ZZ1 ZZ2;  ZZZ12 = ZZZ1 ZZZ2
  const uint32_t m2 = row_mul(row_sel(P, R, a.ZZ, a.ZZZ, q), row_sel(P, R, b.ZZ, b.ZZZ, q), k);
  const uint32_t PP = from_row<0>(m2, l), RR = from_row<1>(m2, l), ZZ12 = from_row<2>(m2, l), ZZZ12 = from_row<3>(m2, l);
  // stage 3: PPP = P PP;  Q = U1 PP;  ZZ3 = ZZ12 PP   (row 3 repeats row 0)
  const uint32_t m3 = row_mul(row_sel(P, U1, ZZ12, P, q), PP, k);
  const uint32_t PPP = from_row<0>(m3, l), Q = from_row<1>(m3, l), ZZ3 = from_row<2>(m3, l);
  // X3 = RR - PPP - 2 Q  (lazy: + 3p - PPP + 2 (3p - Q)), two carry passes: limbs <= 2^28 + 1
  uint32_t X3 = RR + (k.kp3 - PPP) + 2u * (k.kp3 - Q);
  X3 = row_norm_pass(row_norm_pass(X3, l), l);
  // stage 4: rows 0-2: Y3 = R (Q - X3) - S1 PPP;  row 3: ZZZ3 = ZZZ12 PPP (+ 0)
  const uint32_t qx = Q + (k.kp12 - X3), ns1 = k.kp3 - S1;                 // fp_sub<12>, fp_neg<3>
  const bool r3 = q == 3u;
  const uint32_t m4 = row_mul2(r3 ? ZZZ12 : R, r3 ? PPP : qx, r3 ? 0u : PPP, r3 ? 0u : ns1, k);
  xyzz_row r;
  r.X = X3; r.Y = from_row<0>(m4, l); r.ZZ = ZZ3; r.ZZZ = from_row<3>(m4, l); r.inf = 0;
  return r;
}

// 2 a: the stages of quad_dbl
__device__ __forceinline__ xyzz_row row_dbl(const xyzz_row& a, const RowK& k) {
  if (a.inf) return a;
  const uint32_t l = k.lane16, q = k.row;
  const uint32_t U = a.Y + a.Y;                                             // limbs < 2^29 + 8
  // stage 1: row 1: XX = X^2;  the others: V = U^2
  const uint32_t s1 = q == 1u ? a.X : U;
  const uint32_t m1 = row_mul(s1, s1, k);
  const uint32_t V = from_row<0>(m1, l), XX = from_row<1>(m1, l);
  const uint32_t M = XX + XX + XX;                                           // limbs < 3 * 2^28 + 12
  // stage 2: row 0: W = U V;  row 1: S = X V;  row 2: MM = M M;  row 3: ZZ3 = V ZZ
  const uint32_t m2 = row_mul(row_sel(U, a.X, M, V, q), row_sel(V, V, M, a.ZZ, q), k);
  const uint32_t W = from_row<0>(m2, l), S = from_row<1>(m2, l), MM = from_row<2>(m2, l), ZZ3 = from_row<3>(m2, l);
  uint32_t X3 = MM + 2u * (k.kp3 - S);
  X3 = row_norm_pass(row_norm_pass(X3, l), l);
  // stage 3: row 1: ZZZ3 = W ZZZ (+ 0);  the others: Y3 = M (S - X3) - W Y
  const uint32_t sx = S + (k.kp12 - X3), ny = k.kp6 - a.Y;                   // fp_sub<12>, fp_neg<6>
  const bool r1 = q == 1u;
  const uint32_t m3 = row_mul2(r1 ? W : M, r1 ? a.ZZZ : sx, r1 ? 0u : W, r1 ? 0u : ny, k);
  xyzz_row r;
  r.X = X3; r.Y = from_row<0>(m3, l); r.ZZ = ZZ3; r.ZZZ = from_row<1>(m3, l); r.inf = 0;
  return r;
}

// lane l of every row <- word l of a 14-limb coordinate in memory
__device__ __forceinline__ uint32_t row_load14(const uint32_t* limbs, uint32_t lane16) { return lane16 < (uint32_t)NL ? limbs[lane16] : 0u; }

__device__ __forceinline__ xyzz_row row_load_sum(const PointSum* src, uint32_t lane16) {
  xyzz_row r;
  r.X = row_load14(src->c[0], lane16); r.Y = row_load14(src->c[1], lane16); r.ZZ = row_load14(src->c[2], lane16); r.ZZZ = row_load14(src->c[3], lane16);
  r.inf = src->inf;
  return r;
}

// the inverse of row_load_sum: row 0 of the wave writes the record (limbs nearly normal: only row code reads it back)
__device__ __forceinline__ void row_store_sum(PointSum* dst, const xyzz_row& a, uint32_t lane16) {
  const uint32_t lane = (uint32_t)__lane_id();
  if (lane < (uint32_t)NL) { dst->c[0][lane16] = a.X; dst->c[1][lane16] = a.Y; dst->c[2][lane16] = a.ZZ; dst->c[3][lane16] = a.ZZZ; }
  if (lane == 0) dst->inf = a.inf;
}

__device__ __forceinline__ void row_export(const xyzz_row& acc, uint32_t lane16, PointWords* dst) {
  const xyzz res = row_to_xyzz(acc, lane16);
  if ((threadIdx.x & 63u) == 0) {
    xyzz_words o;
    xyzz_export(res, o);
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 12; ++j) dst->w[c][j] = o.w[c][j];
    dst->inf = o.inf;
  }
}

// regime B's per-MSM Horner with one WAVE per MSM: sum_w 2^(c w) S_w over the nwin window sums of MSM j (k_msm_horner_quad's job).  A
// batch of ~1 000 MSMs is one wave per SIMD: the chain of 255 doublings runs at the lone-wave rate of row_dbl instead of quad_dbl's.
__global__ void __launch_bounds__(64) k_msm_horner_row(const PointSum* __restrict__ group_sum, PointWords* __restrict__ out, uint32_t M, uint32_t nwin, uint32_t c) {
  const uint32_t j = blockIdx.x;
  if (j >= M) return;
  const RowK k = row_constants();
  xyzz_row acc; acc.X = acc.Y = acc.ZZ = acc.ZZZ = 0; acc.inf = 1;
  for (int w = (int)nwin - 1; w >= 0; --w) {
    for (uint32_t d = 0; d < c; ++d) acc = row_dbl(acc, k);
    acc = row_add(acc, row_load_sum(group_sum + (size_t)j * nwin + w, k.lane16), k);
  }
  row_export(acc, k.lane16, out + j);
}

// One item of regime A's 2-D bucket reduction (k_small_tree_quad's job: T0 = sum of all row sums, then one masked sum per row / column bit)
// by a block of W <= 16 waves with one limb per lane: wave v adds the selected elements of rank v, v + W, ... one after the other (a
// lone wave's addition is ~1.7 us against a quad's 5-10), an LDS tree joins the waves, wave 0 exports.  256 elements: 8 + 4 dependent
// additions instead of 4 + 4 + 2 quad levels at three to five times the latency each.
// canonical words of a row-form point with FOUR lanes, one coordinate each (row_export: one lane, four conversions in a row)
__device__ __forceinline__ void row_export4(const xyzz_row& racc, uint32_t lane16, PointWords* dst) {
  const xyzz acc = row_to_xyzz(racc, lane16);
  const uint32_t q = threadIdx.x & 63u;
  if (q < 4u) {
    fp coord;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const uint32_t lo = (q & 1u) ? acc.Y.l[j] : acc.X.l[j], hi = (q & 1u) ? acc.ZZZ.l[j] : acc.ZZ.l[j];
      coord.l[j] = (q & 2u) ? hi : lo;
    }
    uint32_t ow[12];
    fp_to_host_words(coord, ow);
    if (acc.inf) {
#pragma unroll
      for (int j = 0; j < 12; ++j) ow[j] = 0;
    }
    uint4* d4 = reinterpret_cast<uint4*>(&dst->w[q][0]);
    d4[0] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
    d4[1] = make_uint4(ow[4], ow[5], ow[6], ow[7]);
    d4[2] = make_uint4(ow[8], ow[9], ow[10], ow[11]);
    if (q == 0u) dst->inf = acc.inf;
  }
}

// out_host != NULL (zero-copy calls): the items go straight into the mapped host records; the LAST block to finish (ticket = status word
// 3, zero at the start of every launch chain) copies the call's status words behind them and publishes the sequence number the host
// polls -- k_export_host's job without its launch.
__global__ void __launch_bounds__(1024) k_small_tree_row(const PointSum* __restrict__ rowsum, const PointSum* __restrict__ colsum,
                                                         PointWords* __restrict__ out, uint32_t hb, uint32_t lb,
                                                         PointWords* __restrict__ out_host, uint32_t* __restrict__ status_words,
                                                         uint32_t* __restrict__ flag_host, uint32_t seq) {
  __shared__ PointSum sh[8];
  const uint32_t item = blockIdx.x, lw = blockIdx.y, wv = threadIdx.x >> 6, W = blockDim.x >> 6;
  const RowK k = row_constants();
  const bool on_rows = item <= hb;
  const uint32_t J = on_rows ? (1u << hb) : (1u << lb);
  const PointSum* src = (on_rows ? rowsum : colsum) + (size_t)lw * J;
  const uint32_t bit = on_rows ? item - 1u : item - 1u - hb;
  xyzz_row acc; acc.X = acc.Y = acc.ZZ = acc.ZZZ = 0; acc.inf = 1;
  const uint32_t cnt = item == 0u ? J : J >> 1;                 // item 0: every element; a bit's item: the elements with that bit set
  for (uint32_t r = wv; r < cnt; r += W) {                      // (wave-uniform trip count)
    const uint32_t e = item == 0u ? r : (((((r >> bit) << 1) | 1u) << bit) | (r & ((1u << bit) - 1u)));
    acc = row_add(acc, row_load_sum(src + e, k.lane16), k);
  }
  for (uint32_t d = W >> 1; d >= 1u; d >>= 1) {
    if (wv >= d && wv < 2u * d) row_store_sum(&sh[wv - d], acc, k.lane16);
    __syncthreads();
    if (wv < d) acc = row_add(acc, row_load_sum(&sh[wv], k.lane16), k);
    __syncthreads();
  }
  PointWords* dst = (out_host ? out_host : out) + (size_t)lw * gridDim.x + item;
  if (wv != 0u) return;                                         // (past the last barrier)
  row_export4(acc, k.lane16, dst);
  if (out_host) {
    __threadfence_system();                                     // the four exporting lanes' stores are visible to the host before the ticket is drawn
    if (threadIdx.x == 0) {
      const uint32_t total = gridDim.x * gridDim.y;
      if (atomicAdd(&status_words[3], 1u) == total - 1u) {
        uint32_t* st = reinterpret_cast<uint32_t*>(out_host + total);
        st[0] = atomicAdd(&status_words[0], 0u); st[1] = atomicAdd(&status_words[1], 0u); st[2] = atomicAdd(&status_words[2], 0u); st[3] = 0;
        __threadfence_system();
        __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
}

// out[i] = addend[i] + scalars[i % nscalars] * bases[i % nbase] with one WAVE per output (k_batch_mul_quad's job for a few hundred to a few
// thousand outputs: the map / fold loops of the callers, curdleproofs.py:310-311, ipa.py:142-146, as a deferred batch): 4-bit windows over
// a table of 15 multiples kept in registers (a point is four VGPRs here), 255 doublings + <= 64 additions.  Inputs affine96 (standard
// form; zeros = identity), output canonical XYZZ words for the host (no inversion on the device).
__global__ void __launch_bounds__(64) k_batch_mul_row(const uint32_t* __restrict__ base_raw, uint32_t nbase, const uint32_t* __restrict__ scalars, uint32_t nscalars,
                                                      const uint32_t* __restrict__ addend_raw, PointWords* __restrict__ out, uint32_t n) {
  const uint32_t i = blockIdx.x;
  if (i >= n) return;
  const RowK k = row_constants();
  const uint32_t l = k.lane16;
  uint32_t r2 = 0;
#pragma unroll
  for (int t = 0; t < NL; ++t) r2 = l == (uint32_t)t ? fp_r2().l[t] : r2;
  auto load_point = [&](const uint32_t* w, xyzz_row& P) {                    // 24 words x | y, standard form -> Montgomery rows
    uint32_t wd[24], any = 0;
    for (int t = 0; t < 24; ++t) { wd[t] = w[t]; any |= wd[t]; }
    P.X = row_mul(row_from_fp(fp_from_words(wd), l), r2, k);
    P.Y = row_mul(row_from_fp(fp_from_words(wd + 12), l), r2, k);
    P.ZZ = k.one; P.ZZZ = k.one; P.inf = any ? 0u : 1u;
  };
  xyzz_row acc; acc.X = acc.Y = acc.ZZ = acc.ZZZ = 0; acc.inf = 1;
  xyzz_row P1;
  load_point(base_raw + 24ull * (i % nbase), P1);
  uint32_t s[8];
  for (int t = 0; t < 8; ++t) s[t] = scalars[8ull * (i % nscalars) + t];
  if (!P1.inf) {
    xyzz_row T[15];                                                          // T[d - 1] = d P1
    T[0] = P1;
    T[1] = row_dbl(P1, k);
#pragma unroll
    for (int d = 2; d < 15; ++d) T[d] = (d & 1) ? row_dbl(T[(d - 1) / 2], k) : row_add(T[d - 1], P1, k);
#pragma unroll 1
    for (int nib = 63; nib >= 0; --nib) {
      acc = row_dbl(row_dbl(row_dbl(row_dbl(acc, k), k), k), k);
      const uint32_t d = (s[nib >> 3] >> ((nib & 7) * 4)) & 15u;            // wave-uniform
      if (d) {
        xyzz_row Td = T[0];
#pragma unroll
        for (int e = 1; e < 15; ++e) {
          const bool me = d == (uint32_t)(e + 1);
          Td.X = me ? T[e].X : Td.X; Td.Y = me ? T[e].Y : Td.Y; Td.ZZ = me ? T[e].ZZ : Td.ZZ; Td.ZZZ = me ? T[e].ZZZ : Td.ZZZ;
          Td.inf = me ? T[e].inf : Td.inf;                                  // (a multiple of a point of small order may be the identity)
        }
        acc = row_add(acc, Td, k);
      }
    }
  }
  if (addend_raw) {
    xyzz_row A;
    load_point(addend_raw + 24ull * i, A);
    acc = row_add(acc, A, k);
  }
  row_export(acc, l, out + i);
}

// flags[i] = 1 iff affine96 point i (standard form, on the curve; zeros = identity) lies in the prime-order subgroup: [z^2] P == phi(P) + P
// (g1_in_subgroup, g1_xyzz.h) with one WAVE per point: 126 doublings + 12 additions of a lone wave, ~0.25 ms for up to ~1 000 points --
// what the deferred G1Point layer asks before it folds a product of products over bases decoded unchecked (py_arkworks_bls12381.py).
__global__ void __launch_bounds__(64) k_subgroup_row(const uint32_t* __restrict__ raw, uint32_t n, uint8_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x;
  if (i >= n) return;
  const RowK k = row_constants();
  const uint32_t l = k.lane16;
  constexpr uint32_t bt[NL] = {D_BETA[0], D_BETA[1], D_BETA[2], D_BETA[3], D_BETA[4], D_BETA[5], D_BETA[6], D_BETA[7], D_BETA[8], D_BETA[9], D_BETA[10], D_BETA[11], D_BETA[12], D_BETA[13]};
  uint32_t r2 = 0, beta = 0;
#pragma unroll
  for (int t = 0; t < NL; ++t) { r2 = l == (uint32_t)t ? fp_r2().l[t] : r2; beta = l == (uint32_t)t ? bt[t] : beta; }
  uint32_t wd[24], any = 0;
  for (int t = 0; t < 24; ++t) { wd[t] = raw[24ull * i + t]; any |= wd[t]; }
  if (!any) { if ((threadIdx.x & 63u) == 0) flags[i] = 1; return; }
  xyzz_row P;
  P.X = row_mul(row_from_fp(fp_from_words(wd), l), r2, k);
  P.Y = row_mul(row_from_fp(fp_from_words(wd + 12), l), r2, k);
  P.ZZ = k.one; P.ZZZ = k.one; P.inf = 0;
  constexpr uint64_t ZABS = 0xd201000000010000ull;
  xyzz_row q = P;
#pragma unroll 1
  for (int bit = 62; bit >= 0; --bit) {
    q = row_dbl(q, k);
    if ((ZABS >> bit) & 1ull) q = row_add(q, P, k);
  }
  xyzz_row acc = q;
#pragma unroll 1
  for (int bit = 62; bit >= 0; --bit) {
    acc = row_dbl(acc, k);
    if ((ZABS >> bit) & 1ull) acc = row_add(acc, q, k);
  }
  xyzz_row N1 = P;                                                             // - P
  N1.Y = row_norm_pass(row_norm_pass(k.kp3 - P.Y, l), l);
  acc = row_add(acc, N1, k);
  xyzz_row N2 = N1;                                                            // - phi(P) = (beta x, -y)
  N2.X = row_mul(P.X, beta, k);
  acc = row_add(acc, N2, k);
  if ((threadIdx.x & 63u) == 0) flags[i] = acc.inf ? 1 : 0;
}

// ---- the measurement behind "does one limb per lane shorten a lone wave's chain of additions?": every wave of the launch runs `iters`
// DEPENDENT additions acc += (P0, P1 alternating).  MODE 0: the one-lane formulas (all 64 lanes compute the same thing: a lone LANE's
// latency); 1: a DPP quad per addition (g1_quad.h); 2: one limb per lane (row_add).  Wave w exports its result as canonical words.
template <int MODE>
__global__ void __launch_bounds__(64) k_probe_add_chain(const PreparedPoint* __restrict__ pts, PointWords* __restrict__ out, int iters) {
  // operands that differ from lane to lane (mode 0) / from quad to quad (mode 1), as in a real kernel: with the same two points in every
  // lane the compiler proves the whole chain wave-uniform and runs it on the SCALAR unit.  Lane 0 / quad 0 / the wave starts from P0.
  const uint32_t swap = MODE == 0 ? (threadIdx.x & 1u) : (MODE == 1 ? ((threadIdx.x >> 2) & 1u) : 0u);
  fp x0, y0, x1, y1;
  uint32_t fl;
  load_affine(pts + swap, x0, y0, fl);
  load_affine(pts + (swap ^ 1u), x1, y1, fl);
  const xyzz p0 = xyzz_from_affine(x0, y0), p1 = xyzz_from_affine(x1, y1);
  xyzz res;
  if (MODE == 2) {
    const RowK k = row_constants();
    const xyzz_row r0 = row_from_xyzz(p0, k.lane16), r1 = row_from_xyzz(p1, k.lane16);
    xyzz_row acc = r0;
    for (int i = 0; i < iters; ++i) acc = row_add(acc, (i & 1) ? r0 : r1, k);
    res = row_to_xyzz(acc, k.lane16);
  } else if (MODE == 1) {
    const uint32_t q = threadIdx.x & 3u;
    xyzz acc = p0;
    for (int i = 0; i < iters; ++i) acc = quad_add(acc, (i & 1) ? p0 : p1, q);
    res = acc;
  } else {
    xyzz acc = p0;
    for (int i = 0; i < iters; ++i) acc = xyzz_add(acc, (i & 1) ? p0 : p1);
    res = acc;
  }
  if (threadIdx.x == 0) {
    xyzz_words o;
    xyzz_export(res, o);
    PointWords* dst = out + blockIdx.x;
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 12; ++j) dst->w[c][j] = o.w[c][j];
    dst->inf = o.inf;
  }
}
