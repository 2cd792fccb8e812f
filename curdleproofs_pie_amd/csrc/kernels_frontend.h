// The shuffle verifier's FRONT-END on the device (SURVEY 8(f) rows 1 + 3): everything csrc/shuffle_verify.cpp's
// prepare_one does per proof between the wire bytes and the row builder's input block -- the Fiat-Shamir transcript
// (curdleproofs_transcript.py:15-25 over merlin_transcripts/{merlin_transcript,strobe,keccak}.py), the grand-product scalar
// (same_perm.py:98-101), D = B - beta^-1 G_sum + alpha H_sum (grand_prod.py:157) and A' = A + T_1 + U_1 (curdleproofs.py:210)
// with their 48-byte encodings, inner_prod (grand_prod.py:164-166) and the challenge inverses (ipa.py:178, same_msm.py:175) --
// one proof per lane, on the wire points the decompression stage already put into HBM.  Part of the single translation unit
// csrc/msm_gpu.hip.
//
// The transcript is cg1merlin::Machine (kernels_merlin.h): a resumable byte-level STROBE state machine per lane, the lanes of a
// wave permuting together.  The program is the verifier's own operation sequence (built on the host by fe_build_program in
// msm_gpu.hip, the mirror of prepare_one); challenges are squeezed straight into their slots of the lane's row-input block
// (cg1rows::RowIn layout, so k_shuffle_rows reads what this kernel wrote).  Three operations are BARRIERS: a lane that reaches
// one waits until every lane of its wave has, then all take the step together -- the field / group arithmetic of a step is
// thousands of instructions and must not run once per straggler:
//   X_GPROD   prod_i (a_i + i alpha + beta)                                          -> scratch (absorbed next)
//   X_DA      beta^-1, inner_prod; D (64 table additions: 8-bit fixed-base tables of G_sum and H_sum, + B), A'; one shared
//             inversion; both compressed                                             -> row-input block + scratch
//   X_FINAL   inverses of the 2 lg round challenges (Montgomery's trick); the finished block is copied to the caller's buffer
// Output = exactly what cg1_shuffle_prepare_inputs writes on the host: the row-input block and the front-end status per proof
// (tests/test_shuffle_frontend_gpu.py compares them byte for byte on every golden proof and tampered variant).
#pragma once
#include "fr.h"

namespace cg1fe {
using cg1fr::fr;
using cg1merlin::COp;
using cg1merlin::LANES;
using cg1merlin::Machine;

enum : uint8_t { X_GPROD = 16, X_DA = 17, X_FINAL = 18 };

struct Params {
  uint32_t ell, lg, L, K;                 // own points per proof, scalars of a row-input block
  uint32_t out_stride;                    // bytes of a lane's private out row: (K + 6) * 32  [block | gprod | r_p | D48 (64 B) | A'48 (64 B)]
  uint32_t idx_A, idx_T1, idx_U1, idx_B, idx_T0;   // own-point indices (csrc/shuffle_verify.cpp Layout)
  uint32_t prio;                          // wave priority (s_setprio): a front-end wave is one long dependent chain sharing its SIMD with throughput kernels
};

__device__ __noinline__ fr fmul(const fr& a, const fr& b) { return cg1fr::fr_mul(a, b); }
__device__ __noinline__ fr fld(const uint8_t* base, uint32_t slot) { fr v; cg1fr::fr_from_le32(base + 32u * slot, v); return v; }
__device__ __noinline__ void fst(uint8_t* base, uint32_t slot, const fr& v) { cg1fr::fr_to_le32(v, base + 32u * slot); }
__device__ inline fr fpow(fr base, uint64_t e) {
  fr acc = cg1fr::fr_one();
  while (e) { if (e & 1) acc = fmul(acc, base); base = fmul(base, base); e >>= 1; }
  return acc;
}
__device__ __noinline__ fr finv(const fr& a) {                  // a^(r-2)
  fr acc = cg1fr::fr_one();
  for (int i = 254; i >= 0; --i) {
    acc = fmul(acc, acc);
    if ((cg1::H_FR_MINUS_2[i >> 6] >> (i & 63)) & 1) acc = fmul(acc, a);
  }
  return acc;
}

// affine standard words -> 48-byte ZCash encoding (the same rule as k_batch_compress)
__device__ inline void compress48(const uint32_t w[24], bool inf, uint8_t* o) {
  if (inf) { o[0] = 0xC0; for (int k = 1; k < 48; ++k) o[k] = 0; return; }
  bool is_large = false, decided = false;
  for (int j = 11; j >= 0 && !decided; --j) if (w[12 + j] != cg1::W_P_MINUS_1_HALF[j]) { is_large = w[12 + j] > cg1::W_P_MINUS_1_HALF[j]; decided = true; }
  for (int j = 0; j < 12; ++j) {
    const uint32_t v = w[11 - j];
    o[4 * j] = (uint8_t)(v >> 24); o[4 * j + 1] = (uint8_t)(v >> 16); o[4 * j + 2] = (uint8_t)(v >> 8); o[4 * j + 3] = (uint8_t)v;
  }
  o[0] |= (uint8_t)(0x80 | (is_large ? 0x20 : 0));
}

// The four decoded points a proof's front-end needs as group elements (A, cm_T.T_1, cm_U.T_1, B: affine96 standard form as
// k_batch_decompress wrote them; zeros = identity / failed decode) -> 128-byte Montgomery records, so that X_DA's additions read
// every operand -- table entry or proof point -- the same way.
__global__ void __launch_bounds__(256) k_fe_gather4(const uint32_t* __restrict__ aff, Params pr, uint32_t n, cg1::PreparedPoint* __restrict__ out) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 4u * n) return;
  const uint32_t proof = t >> 2, j = t & 3u;
  const uint32_t idx = j == 0 ? pr.idx_A : (j == 1 ? pr.idx_T1 : (j == 2 ? pr.idx_U1 : pr.idx_B));
  const uint32_t* src = aff + 24ull * ((size_t)proof * pr.L + idx);
  uint32_t w[24], any = 0;
  for (int k = 0; k < 24; ++k) { w[k] = src[k]; any |= w[k]; }
  cg1::PreparedPoint rec;
  const cg1::fp x = cg1::fp_to_mont(cg1::fp_from_words(w)), y = cg1::fp_to_mont(cg1::fp_from_words(w + 12));
  for (int k = 0; k < cg1::NL; ++k) { rec.x[k] = x.l[k]; rec.y[k] = y.l[k]; }
  rec.flags = any ? 0u : 1u;
  rec.pad[0] = rec.pad[1] = rec.pad[2] = 0;
  out[t] = rec;
}

// ---- the barrier steps (all lanes of the wave that are live take them together)
// 32 little-endian bytes of an out-row slot as they stand (a canonical scalar, NOT in Montgomery form) / back
__device__ __forceinline__ fr fld_raw(const uint8_t* base, uint32_t slot) {
  const uint64_t* q = reinterpret_cast<const uint64_t*>(base + 32u * slot);
  return fr{{q[0], q[1], q[2], q[3]}};
}
__device__ __forceinline__ void fst_raw(uint8_t* base, uint32_t slot, const fr& v) {
  uint64_t* q = reinterpret_cast<uint64_t*>(base + 32u * slot);
  q[0] = v.l[0]; q[1] = v.l[1]; q[2] = v.l[2]; q[3] = v.l[3];
}

__device__ __noinline__ void step_gprod(uint8_t* orow, const Params& pr) {
  const cg1rows::RowIn R{pr.ell, pr.lg};
  // prod_i (a_i + i alpha + beta), i alpha + beta built by repeated addition (same_perm.py:98-101).  The factors stay as they are
  // read (no conversion into Montgomery form): a Montgomery product of two plain values is x y / R, so the running product after
  // ell factors is off by R^-(ell-1), and ONE product with R^ell (= the Montgomery form of R^(ell-1)) at the end puts that right --
  // ell + ~12 products instead of 3 ell.
  const fr alpha = fld_raw(orow, (uint32_t)R.head()), beta = fld_raw(orow, (uint32_t)R.head() + 1u);
  fr t = cg1fr::fr_add(beta, alpha);
  fr g = cg1fr::fr_add(fld_raw(orow, (uint32_t)R.a()), beta);
  for (uint32_t i = 1; i < pr.ell; ++i) {
    g = fmul(g, cg1fr::fr_add(fld_raw(orow, (uint32_t)R.a() + i), t));
    t = cg1fr::fr_add(t, alpha);
  }
  const fr r2{{cg1::H_FR_R2[0], cg1::H_FR_R2[1], cg1::H_FR_R2[2], cg1::H_FR_R2[3]}};      // the Montgomery form of R
  fst_raw(orow, pr.K, fmul(g, fpow(r2, pr.ell - 1u)));
}

__device__ __noinline__ void step_da(uint8_t* orow, const Params& pr, const cg1::PreparedPoint* __restrict__ tabG, const cg1::PreparedPoint* __restrict__ tabH,
                                     const cg1::PreparedPoint* __restrict__ four) {
  using namespace cg1;
  const cg1rows::RowIn R{pr.ell, pr.lg};
  const fr alpha_g = fld(orow, (uint32_t)R.head() + 2u), beta_g = fld(orow, (uint32_t)R.head() + 3u);
  const fr beta_inv = finv(beta_g);
  fst(orow, (uint32_t)R.beta_inv(), beta_inv);
  {
    const fr r_p = fld(orow, pr.K + 1u), gprod = fld(orow, pr.K);
    const fr beta_ell = fpow(beta_g, pr.ell);
    const fr ip = cg1fr::fr_sub(cg1fr::fr_add(fmul(r_p, fmul(beta_ell, beta_g)), fmul(gprod, beta_ell)), cg1fr::fr_one());   // grand_prod.py:164-166
    fst(orow, (uint32_t)R.inner_prod(), ip);
  }
  // D = B - beta^-1 G_sum + alpha_g H_sum: one table entry per scalar byte; then A' = A + T_1 + U_1.  ONE addition call site: operand j
  // of 68 is picked by address (tables, then the proof's four points), the accumulator is parked after B and restarted for A'.
  uint8_t s1[32], s2[32];
  cg1fr::fr_to_le32(cg1fr::fr_neg(beta_inv), s1);
  cg1fr::fr_to_le32(alpha_g, s2);
  xyzz acc = xyzz_identity(), Dp = xyzz_identity();
  auto operand = [&](uint32_t j, bool& skip) -> const PreparedPoint* {
    skip = false;
    if (j < 32u) { const uint32_t b = s1[j]; skip = b == 0u; return tabG + (size_t)j * 256u + b; }
    if (j < 64u) { const uint32_t b = s2[j - 32u]; skip = b == 0u; return tabH + (size_t)(j - 32u) * 256u + b; }
    if (j == 64u) return four + 3;                          // B
    return four + (j - 65u);                                // A, T_1, U_1
  };
  // (the record of addition j + 1 is fetched before addition j runs: a table entry is a 128-byte random access)
  fp nx, ny; uint32_t nflags; bool nskip;
  load_affine(operand(0u, nskip), nx, ny, nflags);
#pragma unroll 1
  for (uint32_t j = 0; j < 68u; ++j) {
    const fp x = nx, y = ny;
    const uint32_t flags = nflags;
    const bool skip = nskip;
    if (j + 1u < 68u) load_affine(operand(j + 1u, nskip), nx, ny, nflags);
    if (j == 65u) { Dp = acc; acc = xyzz_identity(); }
    if (!skip && !(flags & 1u)) acc = xyzz_madd(acc, x, y);
  }
  // one inversion for both: 1 / (ZZ_D ZZZ_D ZZ_A ZZZ_A)
  const fp one = fp_one();
  const fp tD = Dp.inf ? one : fp_mul(Dp.ZZ, Dp.ZZZ), tA = acc.inf ? one : fp_mul(acc.ZZ, acc.ZZZ);
  const fp inv = fp_inv(fp_mul(tD, tA));
  const fp iD = fp_mul(inv, tA), iA = fp_mul(inv, tD);     // 1 / (ZZ ZZZ) of each
  uint32_t w[24];
  if (!Dp.inf) { fp_to_words(fp_mul(Dp.X, fp_mul(iD, Dp.ZZZ)), w); fp_to_words(fp_mul(Dp.Y, fp_mul(iD, Dp.ZZ)), w + 12); }
  compress48(w, Dp.inf != 0, orow + 32u * (pr.K + 2u));
  if (!acc.inf) { fp_to_words(fp_mul(acc.X, fp_mul(iA, acc.ZZZ)), w); fp_to_words(fp_mul(acc.Y, fp_mul(iA, acc.ZZ)), w + 12); }
  compress48(w, acc.inf != 0, orow + 32u * (pr.K + 4u));
}

__device__ __noinline__ void step_final(uint8_t* orow, const Params& pr, uint8_t* __restrict__ block_out) {
  const cg1rows::RowIn R{pr.ell, pr.lg};
  // inverses of gamma[0..lg) and gamma_m[0..lg) with one inversion (fr_batch_inv of the host front-end, per vector there; the
  // inverse of a field element is unique, so sharing the inversion across both vectors gives the same bytes).  Values are read and
  // written as they stand in the block (plain, not Montgomery): with mm(x, y) = x y / R,  mm(plain, Montgomery) is plain and
  // mm(Montgomery, Montgomery) is Montgomery, so only the gammas are converted (one product each, twice).
  const uint32_t m = 2u * pr.lg;
  const fr r2{{cg1::H_FR_R2[0], cg1::H_FR_R2[1], cg1::H_FR_R2[2], cg1::H_FR_R2[3]}};
  fr acc = fr{{1, 0, 0, 0}};                               // plain prefix products, parked in the inverse slots
  for (uint32_t i = 0; i < m; ++i) {
    fst_raw(orow, (uint32_t)R.gam_inv() + i, acc);
    acc = fmul(acc, fmul(fld_raw(orow, (uint32_t)R.gam() + i), r2));
  }
  fr inv = finv(fmul(acc, r2));                            // Montgomery form of 1 / (g_0 .. g_{m-1})
  for (uint32_t i = m; i-- > 0;) {
    const fr pre = fld_raw(orow, (uint32_t)R.gam_inv() + i);
    fst_raw(orow, (uint32_t)R.gam_inv() + i, fmul(inv, pre));
    inv = fmul(inv, fmul(fld_raw(orow, (uint32_t)R.gam() + i), r2));
  }
  const uint4* __restrict__ src = reinterpret_cast<const uint4*>(orow);
  uint4* __restrict__ dst = reinterpret_cast<uint4*>(block_out);
  uint32_t i = 0;
  for (; i + 8u <= 2u * pr.K; i += 8u) {                   // eight loads in flight per round trip (the block is a lane's own 5.8 KB)
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = src[i + k];
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[i + k] = v[k];
  }
  for (; i < 2u * pr.K; ++i) dst[i] = src[i];
}

// grid = ceil(n / lanes_used) blocks of LANES threads.  aux: per proof r_p c_fin d_fin z_k z_t z_u x_fin | rho[12] (19 x 32 bytes, as they
// stand in the proof / as the caller drew them).  scratch: n x out_stride bytes.  rowin: n x K x 32.  status: n front-end codes.
__global__ void __launch_bounds__(LANES) k_shuffle_front_end(const uint8_t* __restrict__ init_state, const COp* __restrict__ ops, uint32_t nops,
                                                             const uint32_t* __restrict__ label_table, uint32_t nlabels, const uint8_t* __restrict__ consts,
                                                             const uint8_t* __restrict__ wire, const uint8_t* __restrict__ aux,
                                                             const cg1::PreparedPoint* __restrict__ four, const cg1::PreparedPoint* __restrict__ tabG,
                                                             const cg1::PreparedPoint* __restrict__ tabH, Params pr, uint8_t* __restrict__ scratch,
                                                             uint8_t* __restrict__ rowin, int32_t* __restrict__ status, uint32_t n, uint32_t lanes_used) {
  __shared__ uint32_t lds[52 * LANES];
  __shared__ uint32_t lds_drawn[8 * LANES];
  __shared__ uint32_t lds_labels[cg1merlin::MAX_LABELS * 8];
  if (pr.prio == 1u) __builtin_amdgcn_s_setprio(1);
  else if (pr.prio == 2u) __builtin_amdgcn_s_setprio(2);
  else if (pr.prio == 3u) __builtin_amdgcn_s_setprio(3);
  const uint32_t t = blockIdx.x * lanes_used + threadIdx.x;
  const bool live = threadIdx.x < lanes_used && t < n;
  for (uint32_t j = threadIdx.x; j < nlabels * 8u; j += LANES) lds_labels[j] = label_table[j];
  Machine m;
  m.w = lds + threadIdx.x;
  m.drawn = lds_drawn + threadIdx.x;
  m.labels = lds_labels;
  m.consts = consts;
  for (int i = 0; i < 50; ++i) m.w[i * LANES] = reinterpret_cast<const uint32_t*>(init_state)[i];
  m.pos = init_state[200]; m.pos_begin = init_state[201]; m.cur_flags = init_state[202];
  m.k = 0; m.ph = 0; m.i = 0; m.hdr = 0; m.stage = 0; m.k_loaded = 0xffffffffu; m.rec = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const size_t me = live ? t : 0;
  const uint8_t* row = wire + me * (size_t)pr.L * 48u;
  uint8_t* orow = scratch + me * (size_t)pr.out_stride;
  bool done = !live;
  if (live) {
    // parse: the proof's Fr fields and the weights must be canonical (Scalar.from_le_bytes raises, util.py:149-153); vec_T[0] must
    // not be the identity (curdleproofs.py:173-174).  Same precedence as prepare_one: scalar, T[0], weight.
    const cg1rows::RowIn R{pr.ell, pr.lg};
    const uint8_t* a = aux + me * (19u * 32u);
    int32_t st = 0;
    for (uint32_t k = 0; k < 19u; ++k) {
      uint64_t v[4];
      for (int q = 0; q < 4; ++q) { uint64_t x = 0; for (int b = 7; b >= 0; --b) x = (x << 8) | a[32u * k + 8u * q + b]; v[q] = x; }
      const bool ok = !cg1fr::geq_r(v);
      if (!ok && k < 7u && st != CG1_SHUFFLE_BAD_SCALAR) st = CG1_SHUFFLE_BAD_SCALAR;
      if (!ok && k >= 7u && st == 0) st = CG1_SHUFFLE_BAD_WEIGHT;
      const uint32_t slot = k == 0u ? pr.K + 1u : (k < 7u ? (uint32_t)R.fields() + (k - 1u) : (uint32_t)R.rho() + (k - 7u));
      for (int q = 0; q < 4; ++q) reinterpret_cast<uint64_t*>(orow + 32u * slot)[q] = v[q];
    }
    if (st != CG1_SHUFFLE_BAD_SCALAR && (row[(size_t)pr.idx_T0 * 48u] & 0x40u)) st = CG1_SHUFFLE_T0_INFINITY;
    status[t] = st;
    if (st) {                                               // rejected before any hashing, like the host front-end: an all-zero block
      uint4* dst = reinterpret_cast<uint4*>(rowin + (size_t)t * pr.K * 32u);
      for (uint32_t i = 0; i < 2u * pr.K; ++i) dst[i] = make_uint4(0, 0, 0, 0);
      done = true;
    }
  }
  for (;;) {
    bool needf = false, blocked = false;
    if (!done) needf = m.advance(ops, nops, row, orow, done, blocked);
    if (__ballot(needf) != 0ull) {
      if (needf) { cg1merlin::keccak_words(m.w); m.pos = 0; m.pos_begin = 0; }
      continue;
    }
    if (__ballot(blocked) == 0ull) break;                   // nobody hashes, nobody waits at a step: every lane is done
    // every lane that is not done stands at the same barrier operation (same program, and nobody passes a barrier alone)
    if (blocked) {
      const uint32_t kind = m.rec.x & 0xffu;
      if (kind == X_GPROD) step_gprod(orow, pr);
      else if (kind == X_DA) step_da(orow, pr, tabG, tabH, four + 4u * me);
      else step_final(orow, pr, rowin + (size_t)t * pr.K * 32u);
      ++m.k; m.ph = 0; m.i = 0; m.stage = 0;
    }
  }
}


// ------------------------------------------------------------------ the same front-end over a BLOCK PROGRAM (round 3, second form)
// k_shuffle_front_end above spends 2/3 of a pass outside Keccak-f: its lanes stand at different byte positions after their first
// rejected challenge draws, so the STROBE byte machine runs as the union of the lanes' phases, and every message is an exposed
// global load.  But the verifier's transcript has a STATIC shape: every message length is fixed by ell, and a challenge always
// leaves the sponge at (pos, pos_begin) = (32, 0) however many draws it took (the PRF's C flag forces a permutation before each
// draw) -- so the whole byte stream between two permutations is known per NODE of a small control-flow graph:
//     plain node -> next node;     squeeze node -accepted-> next node,  -rejected-> its redo node (frame + PRF header at pos 32) -> itself / next
// The host cuts the operation list into those nodes once per ell (build_block_program in msm_gpu.hip: a symbolic run of strobe.py:55-107
// and merlin_transcript.py:11-24 that records, per sponge byte, the constant XOR-ed into it or where the byte comes from);
// cg1merlin::k_fill_rows then writes, for every proof, one 192-byte ROW per node: 42 words = everything XOR-ed into the rate between two
// permutations that is known before hashing starts (framing, labels, lengths, STROBE's own marks, the proof's and the instance's
// point encodings -- canonicalised where flagged as infinity, util.py:27-32), 1 word of node information and 5 piece descriptors
// for what is not (re-appended challenges, the grand product, D, inner_prod, A': copied from the lane's LDS / out row when the lane
// gets there).  A pass of the hashing kernel is then: XOR the pieces, issue the loads of BOTH successor rows, Keccak-f with the
// current row folded into its load of the sponge, pick the successor.  No byte machine, no exposed latency, the same code for
// every lane whatever node it stands at.  Same outputs, byte for byte (tests/test_shuffle_frontend_gpu.py runs both forms).
using cg1merlin::ROW_WORDS; using cg1merlin::N_PLAIN; using cg1merlin::N_SQUEEZE; using cg1merlin::N_END; using cg1merlin::MAX_PIECES;
using cg1merlin::RowDesc; using cg1merlin::row_word_index; using cg1merlin::apply_piece; using cg1merlin::keccak_absorb_row; using cg1merlin::load_row;

template <bool TIMED>
__global__ void __launch_bounds__(LANES) k_shuffle_front_end_rows(const uint8_t* __restrict__ init_state, const uint32_t* __restrict__ rows, uint32_t nodes,
                                                                  const uint8_t* __restrict__ wire, const uint8_t* __restrict__ aux,
                                                                  const cg1::PreparedPoint* __restrict__ four, const cg1::PreparedPoint* __restrict__ tabG,
                                                                  const cg1::PreparedPoint* __restrict__ tabH, Params pr, uint8_t* __restrict__ scratch,
                                                                  uint8_t* __restrict__ rowin, int32_t* __restrict__ status, uint32_t n, uint32_t lanes_used,
                                                                  uint32_t* __restrict__ passes_out) {
  __shared__ uint32_t lds[52 * LANES];
  __shared__ uint32_t lds_drawn[9 * LANES];                  // (+1: the word-wise copy reads one word past the challenge)
  const uint32_t t = blockIdx.x * lanes_used + threadIdx.x;
  const bool live = threadIdx.x < lanes_used && t < n;
  uint32_t* w = lds + threadIdx.x;
  uint32_t* drawn = lds_drawn + threadIdx.x;
  drawn[8 * LANES] = 0u;
  for (int i = 0; i < 50; ++i) w[i * LANES] = reinterpret_cast<const uint32_t*>(init_state)[i];
  const size_t me = live ? t : 0;
  const uint8_t* row = wire + me * (size_t)pr.L * 48u;
  uint8_t* orow = scratch + me * (size_t)pr.out_stride;
  bool done = !live;
  if (live) {                                               // the same parse as k_shuffle_front_end
    const cg1rows::RowIn R{pr.ell, pr.lg};
    const uint8_t* a = aux + me * (19u * 32u);
    int32_t st = 0;
    for (uint32_t k = 0; k < 19u; ++k) {
      uint64_t v[4];
      for (int q = 0; q < 4; ++q) { uint64_t x = 0; for (int b = 7; b >= 0; --b) x = (x << 8) | a[32u * k + 8u * q + b]; v[q] = x; }
      const bool ok = !cg1fr::geq_r(v);
      if (!ok && k < 7u && st != CG1_SHUFFLE_BAD_SCALAR) st = CG1_SHUFFLE_BAD_SCALAR;
      if (!ok && k >= 7u && st == 0) st = CG1_SHUFFLE_BAD_WEIGHT;
      const uint32_t slot = k == 0u ? pr.K + 1u : (k < 7u ? (uint32_t)R.fields() + (k - 1u) : (uint32_t)R.rho() + (k - 7u));
      for (int q = 0; q < 4; ++q) reinterpret_cast<uint64_t*>(orow + 32u * slot)[q] = v[q];
    }
    if (st != CG1_SHUFFLE_BAD_SCALAR && (row[(size_t)pr.idx_T0 * 48u] & 0x40u)) st = CG1_SHUFFLE_T0_INFINITY;
    status[t] = st;
    if (st) {
      uint4* dst = reinterpret_cast<uint4*>(rowin + (size_t)t * pr.K * 32u);
      for (uint32_t i = 0; i < 2u * pr.K; ++i) dst[i] = make_uint4(0, 0, 0, 0);
      done = true;
    }
  }
  // row of node k of this lane: my_rows + k * row_step (layout [wave][node][lane][48 words], see cg1merlin::k_fill_rows)
  const uint4* my_rows = reinterpret_cast<const uint4*>(rows + row_word_index((uint32_t)me, 0u, nodes, lanes_used));
  const size_t row_step = (size_t)lanes_used * (ROW_WORDS / 4);
  uint32_t cur[ROW_WORDS], nxa[ROW_WORDS], nxr[ROW_WORDS];
  load_row(my_rows, cur);
  uint32_t nd = 0, passes = 0;
  bool bar_done = false;
  unsigned long long t_pieces = 0, t_kec = 0, t_post = 0, t_step[3] = {0, 0, 0}, t_sq = 0;      // TIMED: shader clock per part of a pass (reported per wave)
  auto now = [&]() -> unsigned long long { return TIMED ? __builtin_amdgcn_s_memtime() : 0ull; };
  for (;;) {
    const unsigned long long c0 = now();
    const uint32_t info = cur[42];
    const uint32_t type = info & 3u, bar = (info >> 2) & 3u;
    const bool blocked = !done && bar != 0u && !bar_done;
    const bool act = !done && !blocked;
    if (__ballot(act) != 0ull) {
      if (act) {
        if (type == N_END) {
          done = true;
        } else {
          // what was not known before hashing started: the accepted challenge (LDS) or values of the lane's out row
#pragma unroll
          for (uint32_t q = 0; q < MAX_PIECES; ++q) {
            const uint32_t pc = cur[43 + q];
            const uint32_t len = pc & 63u;
            if (len != 0u) {
              apply_piece(pc, w, drawn, orow);
            }
          }
          const uint32_t da = (info >> 4) & 3u, dr = (info >> 6) & 3u;
          load_row(my_rows + (size_t)(nd + da) * row_step, nxa);
          if (type == N_SQUEEZE) load_row(my_rows + (size_t)(nd + dr) * row_step, nxr);
          const unsigned long long c1 = now();
          keccak_absorb_row(w, cur);
          const unsigned long long c2 = now();
          t_pieces += c1 - c0; t_kec += c2 - c1;
          bool accept = true;
          uint32_t dv[8];
          if (type == N_SQUEEZE) {                                            // strobe.py:77-87 from pos 0, curdleproofs_transcript.py:15-25
            uint64_t wv[4], any = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { dv[j] = w[j * LANES]; w[j * LANES] = 0u; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { wv[j] = (uint64_t)dv[2 * j] | ((uint64_t)dv[2 * j + 1] << 32); any |= wv[j]; }
            accept = any != 0;
            if (accept) { accept = false; for (int j = 3; j >= 0; --j) if (wv[j] != cg1::H_FR[j]) { accept = wv[j] < cg1::H_FR[j]; break; } }
            if (accept) {
#pragma unroll
              for (int j = 0; j < 8; ++j) drawn[j * LANES] = dv[j];
            }
          }
          nd += accept ? da : dr;
          t_sq += now() - c2;
#pragma unroll
          for (uint32_t j = 0; j < ROW_WORDS; ++j) cur[j] = accept ? nxa[j] : nxr[j];
          // the accepted challenge goes to its slot of the out row AFTER the successor row has been waited for: a store issued before
          // that wait would be waited for as well (one counter for loads and stores), ~5 K clocks per pass
          asm volatile("" ::: "memory");
          if (type == N_SQUEEZE && accept) {
            uint32_t* o = reinterpret_cast<uint32_t*>(orow + 32u * ((info >> 8) & 0xffffu));
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = dv[j];
          }
          bar_done = false;
        }
      }
      ++passes;
      t_post += now() - c0;
      continue;
    }
    if (__ballot(blocked) == 0ull) break;
    if (blocked) {                                            // every lane that is not done stands at the same barrier
      if (bar == 1u) step_gprod(orow, pr);
      else if (bar == 2u) step_da(orow, pr, tabG, tabH, four + 4u * me);
      else step_final(orow, pr, rowin + (size_t)t * pr.K * 32u);
      bar_done = true;
    }
    if (TIMED) { const unsigned long long dt = now() - c0; const uint32_t b = __builtin_amdgcn_readfirstlane(blocked ? bar : 0u); if (b) t_step[b - 1u] += dt; }
  }
  if (passes_out && threadIdx.x == 0) {                     // passes | clocks / 256: pieces + row loads issued, Keccak-f, whole passes, draw + range check, the three barrier steps
    passes_out[8 * blockIdx.x] = passes;
    passes_out[8 * blockIdx.x + 1] = (uint32_t)(t_pieces >> 8); passes_out[8 * blockIdx.x + 2] = (uint32_t)(t_kec >> 8);
    passes_out[8 * blockIdx.x + 3] = (uint32_t)(t_post >> 8); passes_out[8 * blockIdx.x + 4] = (uint32_t)(t_sq >> 8);
    for (int k = 0; k < 3; ++k) passes_out[8 * blockIdx.x + 5 + k] = (uint32_t)(t_step[k] >> 8);
  }
}

}  // namespace cg1fe
