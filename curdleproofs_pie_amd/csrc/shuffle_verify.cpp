// Batch verifier front-end for the curdleproofs shuffle argument: turns B serialized
// proofs + their instances into ONE multi-scalar-multiplication identity check for the GPU.
//
// The reference verifies one proof at a time in Python (curdleproofs.py:160-246 -> same_perm.py:75-121 ->
// grand_prod.py:161-218 -> ipa.py:190-236, same_scalar.py:71-111, same_msm.py:184-227), doing every scalar
// operation on a `Scalar` object, every left-hand side with G1 operators, and batching only the right-hand
// sides in MSMAccumulator (msm_accumulator.py:37-68).  This file restates the verifier's EQUATIONS batch-first:
//
//   * the wire bytes are parsed in place (no G1Point objects): WhiskShuffleProof.from_bytes
//     (whisk_interface.py:64-69) = M | CurdleProofsProof (curdleproofs.py:270-281) with the sub-proof layouts of
//     same_perm.py:140-146, grand_prod.py:200-207, ipa.py:271-284, same_scalar.py:141-149, same_msm.py:270-285;
//   * the Fiat-Shamir transcript (native Merlin, csrc/merlin.cpp) absorbs the 48-byte encodings straight from
//     the wire -- a valid encoding is canonical, so compress(decompress(x)) == x -- and only the two points the
//     verifier itself derives, D (grand_prod.py:186) and A' (curdleproofs.py:204), are computed on the host;
//   * every check -- the eight accumulate_check calls AND the four same-scalar equalities the reference asserts
//     directly (same_scalar.py:108) -- is expanded to   sum_k scalar_k * point_k = 0   over the points AS THEY
//     APPEAR ON THE WIRE (left-hand sides included: D and A' are expanded into their summands), each check
//     weighted by its own caller-supplied random rho (the role of msm_accumulator.py:43);
//   * output per proof: a fixed-layout scalar vector over its own 4*ell + 19 + 10*lg wire points and one over the
//     ell + 9 CRS points.  CRS scalars of many proofs add up, so a batch is ONE regime-A MSM (or independent
//     regime-B MSMs when a batch fails and the culprit must be found).
// Point decompression of everything but {A, T_1, U_1, B} and the MSM itself run on the GPU
// (cg1_batch_decompress_device, cg1_msm_device / cg1_msm_batched_device); this file is host logic only.
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/curdle_g1.h"
#include "fr.h"
#include "host_g1.h"
#include "merlin_group.h"
#include "pool.h"

using namespace cg1fr;
using cg1h::jac;

namespace {

constexpr size_t NB = 4;            // N_BLINDERS, curdleproofs.py:26 (the reference hard-codes the 2 + 2 split)

struct Aff { cg1h::fe x, y; bool inf; };

// 8-bit windows over affine entries: tab[w][d-1] = d * 256^w * P  (32 mixed additions per scalar multiplication)
struct FixedBase {
  std::vector<Aff> tab;
  void build(const jac& P) {
    std::vector<jac> t(32 * 255);
    jac base = P;
    for (int w = 0; w < 32; ++w) {
      jac acc = base;
      for (int d = 1; d <= 255; ++d) {
        t[(size_t)w * 255 + (d - 1)] = acc;
        acc = cg1h::jac_add(acc, base);
      }
      base = acc;                   // 256 * base
    }
    std::vector<cg1h::fe> xs(t.size()), ys(t.size());
    std::vector<uint8_t> inf(t.size());
    cg1h::jac_batch_to_affine(t.data(), t.size(), xs.data(), ys.data(), inf.data());
    tab.resize(t.size());
    for (size_t i = 0; i < t.size(); ++i) tab[i] = Aff{xs[i], ys[i], inf[i] != 0};
  }
  void mul_into(jac& acc, const fr& k) const {       // acc += k * P
    uint8_t le[32];
    fr_to_le32(k, le);
    for (int w = 0; w < 32; ++w) {
      if (!le[w]) continue;
      const Aff& e = tab[(size_t)w * 255 + (le[w] - 1)];
      if (!e.inf) acc = cg1h::jac_madd(acc, e.x, e.y);
    }
  }
};

// A wire point as an affine group element: from the GPU's decompression (affine96: x || y little-endian, all-zero =
// identity) when the caller has it, else by a host square root.  false = the encoding is invalid.
bool decode_point(const uint8_t* wire48, const uint8_t* decoded96, Aff& out) {
  if (decoded96) {
    bool zero = true;
    for (int i = 0; i < 96; ++i) zero = zero && decoded96[i] == 0;
    out.inf = zero;
    if (zero) { out.x = cg1h::fe_zero(); out.y = cg1h::fe_zero(); return true; }
    return cg1h::fe_from_le48(decoded96, out.x) && cg1h::fe_from_le48(decoded96 + 48, out.y);
  }
  jac t;
  if (cg1h::g1_decompress(wire48, false, t)) return false;
  out.inf = cg1h::jac_is_identity(t);
  out.x = t.X;                      // jac_from_affine leaves Z = 1
  out.y = t.Y;
  return true;
}
inline void madd_aff(jac& acc, const Aff& p) { if (!p.inf) acc = cg1h::jac_madd(acc, p.x, p.y); }

struct Crs {
  size_t ell = 0, lg = 0;
  std::vector<uint8_t> bytes;       // (ell + 9) * 48, CurdleproofsCrs.to_bytes order (crs.py:92-101)
  FixedBase g_sum, h_sum;
  const uint8_t* H48() const { return bytes.data() + (ell + NB) * 48; }
};

// positions of a proof's own points (the scalar-vector layout written to out_scalars)
struct Layout {
  size_t ell, lg;
  explicit Layout(size_t e, size_t l) : ell(e), lg(l) {}
  size_t R(size_t i) const { return i; }
  size_t S(size_t i) const { return ell + i; }
  size_t T(size_t i) const { return 2 * ell + i; }
  size_t U(size_t i) const { return 3 * ell + i; }
  size_t base() const { return 4 * ell; }
  // wire order of the proof's points (Fr fields skipped)
  size_t M() const { return base() + 0; }
  size_t A() const { return base() + 1; }
  size_t T1() const { return base() + 2; }
  size_t T2() const { return base() + 3; }
  size_t U1() const { return base() + 4; }
  size_t U2() const { return base() + 5; }
  size_t Rp() const { return base() + 6; }
  size_t Sp() const { return base() + 7; }
  size_t B() const { return base() + 8; }
  size_t C() const { return base() + 9; }
  size_t Bc() const { return base() + 10; }
  size_t Bd() const { return base() + 11; }
  size_t LC(size_t j) const { return base() + 12 + j; }
  size_t RC(size_t j) const { return base() + 12 + lg + j; }
  size_t LD(size_t j) const { return base() + 12 + 2 * lg + j; }
  size_t RD(size_t j) const { return base() + 12 + 3 * lg + j; }
  size_t cmA1() const { return base() + 12 + 4 * lg; }
  size_t cmA2() const { return cmA1() + 1; }
  size_t cmB1() const { return cmA1() + 2; }
  size_t cmB2() const { return cmA1() + 3; }
  size_t Ba() const { return cmA1() + 4; }
  size_t Bt() const { return cmA1() + 5; }
  size_t Bu() const { return cmA1() + 6; }
  size_t LA(size_t j) const { return cmA1() + 7 + j; }
  size_t LT(size_t j) const { return cmA1() + 7 + lg + j; }
  size_t LU(size_t j) const { return cmA1() + 7 + 2 * lg + j; }
  size_t RA(size_t j) const { return cmA1() + 7 + 3 * lg + j; }
  size_t RT(size_t j) const { return cmA1() + 7 + 4 * lg + j; }
  size_t RU(size_t j) const { return cmA1() + 7 + 5 * lg + j; }
  size_t count() const { return 4 * ell + 19 + 10 * lg; }
  // CRS scalar-vector layout (crs.py:92-101)
  size_t cG(size_t i) const { return i; }
  size_t cHv(size_t i) const { return ell + i; }
  size_t cH() const { return ell + NB; }
  size_t cGt() const { return ell + NB + 1; }
  size_t cGu() const { return ell + NB + 2; }
  size_t cGsum() const { return ell + NB + 3; }
  size_t cHsum() const { return ell + NB + 4; }
  size_t ncrs() const { return ell + NB + 5; }
};

size_t proof_wire_bytes(size_t lg) { return 48 * (19 + 10 * lg) + 32 * 7; }

// The wheel decodes ANY encoding with the infinity flag set as the identity (host_g1.cpp g1_decompress) and the reference hashes
// points after re-serialising them (points_projective_to_bytes, util.py:27-32): what enters the transcript for such a point is
// the canonical 0xC0 00 .. 00, not the wire bytes.  Applied to the gathered copy of the wire points before any hashing.
inline void canonicalize_infinities(uint8_t* pts48, size_t count) {
  for (size_t i = 0; i < count; ++i) {
    uint8_t* b = pts48 + 48 * i;
    if ((b[0] & 0xC0) == 0xC0) { memset(b, 0, 48); b[0] = 0xC0; }
  }
}

struct Transcript {
  uint8_t st[CG1_MERLIN_STATE_BYTES];
  explicit Transcript(const char* label) { cg1_merlin_init(st, (const uint8_t*)label, strlen(label)); }
  void append(const char* label, const uint8_t* msg, size_t len) { cg1_merlin_append(st, (const uint8_t*)label, strlen(label), msg, len); }
  void point(const char* label, const uint8_t* p48) { append(label, p48, 48); }
  void scalar(const char* label, const fr& s) {
    uint8_t b[32];
    fr_to_le32(s, b);
    append(label, b, 32);
  }
  fr challenge(const char* label) {                      // curdleproofs_transcript.py:15-25
    uint8_t b[32];
    cg1_merlin_challenge_scalar(st, (const uint8_t*)label, strlen(label), b);
    fr s;
    fr_from_le32(b, s);                                  // canonical by construction
    return s;
  }
};

// s_i = prod_{j : bit j of i, MSB first, is set} gamma_j   (util.py:71-78 + ipa.py:179-183 / same_msm.py:176-180)
void fold_scalars(const std::vector<fr>& gamma, std::vector<fr>& s) {
  s.assign(1, fr_one());
  for (const fr& g : gamma) {
    std::vector<fr> nx(s.size() * 2);
    for (size_t k = 0; k < s.size(); ++k) { nx[2 * k] = s[k]; nx[2 * k + 1] = fr_mul(s[k], g); }
    s.swap(nx);
  }
}

using cg1::Pool;   // csrc/pool.h: the persistent worker pool (shared with lazy_host.cpp)

// The per-proof block the DEVICE row builder (csrc/kernels_rows.h, k_shuffle_rows) works from, instead of the rows themselves:
// every challenge the transcript produced plus the handful of scalars derived from them on the host, 32-byte little-endian
// canonical each.  Layout = cg1rows::RowIn:  alpha_p beta_p alpha_g beta_g alpha_i beta_i alpha_s alpha_m | gamma[lg] |
// gamma_m[lg] | a[ell] | beta^-1 | inner_prod | gamma^-1[lg] | gamma_m^-1[lg] | c d z_k z_t z_u x | rho[12].
size_t rowin_scalars(size_t ell, size_t lg) { return 28 + 4 * lg + ell; }
void write_rowin(uint8_t* o, size_t ell, size_t lg, const fr head[8], const fr* gam, const fr* gm, const fr* a, const fr& beta_inv,
                 const fr& inner_prod, const fr* gam_inv, const fr* gm_inv, const fr fields[6], const fr* rho) {
  auto put = [&](const fr& v) { fr_to_le32(v, o); o += 32; };
  for (int k = 0; k < 8; ++k) put(head[k]);
  for (size_t j = 0; j < lg; ++j) put(gam[j]);
  for (size_t j = 0; j < lg; ++j) put(gm[j]);
  for (size_t i = 0; i < ell; ++i) put(a[i]);
  put(beta_inv); put(inner_prod);
  for (size_t j = 0; j < lg; ++j) put(gam_inv[j]);
  for (size_t j = 0; j < lg; ++j) put(gm_inv[j]);
  for (int k = 0; k < 6; ++k) put(fields[k]);
  for (int k = 0; k < 12; ++k) put(rho[k]);
}

// One proof.  Returns 0 (prepared) or a CG1_SHUFFLE_* reject code.
int prepare_one(const Crs& crs, const uint8_t* inst /* 4*ell*48 */, const uint8_t* proof, const uint8_t* weights /* 12*32 */,
                const uint8_t* decoded /* own points 4*ell+1 .. 4*ell+8 (A T_1 T_2 U_1 U_2 R S B) as affine96, or NULL */,
                uint8_t* out_points, uint8_t* out_scalars, uint8_t* out_crs_scalars, uint8_t* out_challenges,
                fr* alpha_s_out = nullptr /* the same-scalar challenge (same_scalar.py:99) */,
                uint8_t* out_rowin = nullptr /* non-NULL: emit the device row builder's input block instead of the rows */) {
  const size_t ell = crs.ell, lg = crs.lg, n = ell + NB;
  const Layout L(ell, lg);

  // ---- gather the wire points into the output layout and the Fr fields into scalars
  memcpy(out_points, inst, 4 * ell * 48);
  fr r_p, c_fin, d_fin, z_k, z_t, z_u, x_fin;
  {
    const uint8_t* p = proof;
    uint8_t* o = out_points + L.base() * 48;
    auto pts = [&](size_t k) { memcpy(o, p, 48 * k); o += 48 * k; p += 48 * k; };
    bool ok = true;
    auto sc = [&](fr& dst) { ok = fr_from_le32(p, dst) && ok; p += 32; };
    pts(10);                                   // M A T_1 T_2 U_1 U_2 R S B C
    sc(r_p);
    pts(2 + 4 * lg);                           // B_c B_d L_C R_C L_D R_D
    sc(c_fin); sc(d_fin);
    pts(4);                                    // cm_A, cm_B
    sc(z_k); sc(z_t); sc(z_u);
    pts(3 + 6 * lg);                           // B_a B_t B_u L_A L_T L_U R_A R_T R_U
    sc(x_fin);
    if (!ok) return CG1_SHUFFLE_BAD_SCALAR;    // Scalar.from_le_bytes raises (util.py:151)
  }
  canonicalize_infinities(out_points, L.count());
  auto P = [&](size_t idx) { return out_points + idx * 48; };
  if (P(L.T(0))[0] & 0x40) return CG1_SHUFFLE_T0_INFINITY;          // curdleproofs.py:173-174

  fr rho[12];
  for (int k = 0; k < 12; ++k)
    if (!fr_from_le32(weights + 32 * k, rho[k])) return CG1_SHUFFLE_BAD_WEIGHT;

  // the four points the verifier needs as group elements
  Aff aA, aT1, aU1, aB;
  auto dec = [&](size_t idx) { return decoded ? decoded + 96 * (idx - L.A()) : nullptr; };
  if (!decode_point(P(L.A()), dec(L.A()), aA) || !decode_point(P(L.T1()), dec(L.T1()), aT1) ||
      !decode_point(P(L.U1()), dec(L.U1()), aU1) || !decode_point(P(L.B()), dec(L.B()), aB))
    return CG1_SHUFFLE_BAD_POINT;

  std::vector<fr> sc(L.count(), fr_zero()), cs(L.ncrs(), fr_zero());
  auto add = [](fr& dst, const fr& v) { dst = fr_add(dst, v); };
  auto sub = [](fr& dst, const fr& v) { dst = fr_sub(dst, v); };

  // ---- curdleproofs.py:176-180
  Transcript tr("curdleproofs");
  for (size_t i = 0; i < 4 * ell; ++i) tr.point("curdleproofs_step1", P(i));
  tr.point("curdleproofs_step1", P(L.M()));
  std::vector<fr> a(ell);
  for (size_t i = 0; i < ell; ++i) a[i] = tr.challenge("curdleproofs_vec_a");

  // ---- same_perm.py:91-109
  tr.point("same_perm_step1", P(L.A()));
  tr.point("same_perm_step1", P(L.M()));
  for (size_t i = 0; i < ell; ++i) tr.scalar("same_perm_step1", a[i]);
  const fr alpha_p = tr.challenge("same_perm_alpha"), beta_p = tr.challenge("same_perm_beta");
  fr gprod = fr_one();
  {
    fr t = beta_p;                                         // i * alpha + beta, built by repeated addition
    for (size_t i = 0; i < ell; ++i) { gprod = fr_mul(gprod, fr_add(a[i], t)); t = fr_add(t, alpha_p); }
  }
  // E1:  B - A - alpha M - sum beta G_i = 0
  add(sc[L.B()], rho[0]); sub(sc[L.A()], rho[0]); sub(sc[L.M()], fr_mul(rho[0], alpha_p));
  {
    const fr t = fr_mul(rho[0], beta_p);
    for (size_t i = 0; i < ell; ++i) sub(cs[L.cG(i)], t);
  }

  // ---- grand_prod.py:175-199
  tr.point("gprod_step1", P(L.B()));
  tr.scalar("gprod_step1", gprod);
  const fr alpha_g = tr.challenge("gprod_alpha");
  tr.point("gprod_step2", P(L.C()));
  tr.scalar("gprod_step2", r_p);
  const fr beta_g = tr.challenge("gprod_beta");
  const fr beta_inv = fr_inv(beta_g);
  std::vector<fr> u(n);
  {
    fr pw = beta_inv;
    for (size_t i = 0; i < ell; ++i) { u[i] = pw; pw = fr_mul(pw, beta_inv); }
    for (size_t i = ell; i < n; ++i) u[i] = pw;          // beta^-(ell+1), n_blinders times
  }
  // D = B - beta^-1 G_sum + alpha H_sum (grand_prod.py:186) and A' = A + T_1 + U_1 (curdleproofs.py:204): the two points
  // the verifier derives itself and feeds to the transcript; normalised together (one field inversion)
  uint8_t D48[48], Ap48[48];
  {
    jac two[2] = {cg1h::jac_identity(), cg1h::jac_identity()};
    crs.g_sum.mul_into(two[0], fr_neg(beta_inv));
    crs.h_sum.mul_into(two[0], alpha_g);
    madd_aff(two[0], aB);
    madd_aff(two[1], aA); madd_aff(two[1], aT1); madd_aff(two[1], aU1);
    cg1h::fe xs[2], ys[2];
    uint8_t inf[2];
    cg1h::jac_batch_to_affine(two, 2, xs, ys, inf);
    cg1h::g1_compress_affine(xs[0], ys[0], inf[0] != 0, D48);
    cg1h::g1_compress_affine(xs[1], ys[1], inf[1] != 0, Ap48);
  }
  const fr beta_ell = fr_pow_u64(beta_g, ell);
  const fr inner_prod = fr_sub(fr_add(fr_mul(r_p, fr_mul(beta_ell, beta_g)), fr_mul(gprod, beta_ell)), fr_one());

  // ---- ipa.py:204-236 (+ :156-186)
  tr.point("ipa_step1", P(L.C()));
  tr.point("ipa_step1", D48);
  tr.scalar("ipa_step1", inner_prod);
  tr.point("ipa_step1", P(L.Bc()));
  tr.point("ipa_step1", P(L.Bd()));
  const fr alpha_i = tr.challenge("ipa_alpha"), beta_i = tr.challenge("ipa_beta");
  std::vector<fr> gam(lg), gam_inv(lg);
  for (size_t j = 0; j < lg; ++j) {
    tr.point("ipa_loop", P(L.LC(j)));
    tr.point("ipa_loop", P(L.LD(j)));
    tr.point("ipa_loop", P(L.RC(j)));
    tr.point("ipa_loop", P(L.RD(j)));
    gam[j] = tr.challenge("ipa_gamma");
  }
  gam_inv = gam;
  fr_batch_inv(gam_inv.data(), lg);
  std::vector<fr> s, s_inv;
  fold_scalars(gam, s);
  fold_scalars(gam_inv, s_inv);
  {
    // E2: sum gamma_j L_C[j] + B_c + alpha C + (alpha^2 ip beta - c d beta) H + sum gamma_j^-1 R_C[j] - sum c s_i G'_i = 0
    const fr w = rho[1];
    for (size_t j = 0; j < lg; ++j) { add(sc[L.LC(j)], fr_mul(w, gam[j])); add(sc[L.RC(j)], fr_mul(w, gam_inv[j])); }
    add(sc[L.Bc()], w);
    add(sc[L.C()], fr_mul(w, alpha_i));
    const fr hcoef = fr_mul(beta_i, fr_sub(fr_mul(fr_sqr(alpha_i), inner_prod), fr_mul(c_fin, d_fin)));
    add(cs[L.cH()], fr_mul(w, hcoef));
    const fr wc = fr_mul(w, c_fin);
    for (size_t i = 0; i < n; ++i) sub(cs[i], fr_mul(wc, s[i]));      // G' = vec_G | vec_H = CRS slots 0..n-1
  }
  {
    // E3: sum gamma_j L_D[j] + B_d + alpha (B - beta^-1 G_sum + alpha_g H_sum) + sum gamma_j^-1 R_D[j] - sum d s_i^-1 u_i G'_i = 0
    const fr w = rho[2];
    for (size_t j = 0; j < lg; ++j) { add(sc[L.LD(j)], fr_mul(w, gam[j])); add(sc[L.RD(j)], fr_mul(w, gam_inv[j])); }
    add(sc[L.Bd()], w);
    const fr wa = fr_mul(w, alpha_i);
    add(sc[L.B()], wa);
    sub(cs[L.cGsum()], fr_mul(wa, beta_inv));
    add(cs[L.cHsum()], fr_mul(wa, alpha_g));
    const fr wd = fr_mul(w, d_fin);
    for (size_t i = 0; i < n; ++i) sub(cs[i], fr_mul(wd, fr_mul(s_inv[i], u[i])));
  }

  // ---- same_scalar.py:82-108
  {
    const size_t order[10] = {L.Rp(), L.Sp(), L.T1(), L.T2(), L.U1(), L.U2(), L.cmA1(), L.cmA2(), L.cmB1(), L.cmB2()};
    for (size_t k = 0; k < 10; ++k) tr.point("sameexp_points", P(order[k]));
  }
  const fr alpha_s = tr.challenge("same_scalar_alpha");
  if (alpha_s_out) *alpha_s_out = alpha_s;
  {
    // z_t G_t = cmA.T_1 + alpha T_1;   z_k R + z_t H = cmA.T_2 + alpha T_2;   same for (u, S, cmB, U)
    const fr w1 = rho[8], w2 = rho[9], w3 = rho[10], w4 = rho[11];
    add(cs[L.cGt()], fr_mul(w1, z_t)); sub(sc[L.cmA1()], w1); sub(sc[L.T1()], fr_mul(w1, alpha_s));
    add(sc[L.Rp()], fr_mul(w2, z_k)); add(cs[L.cH()], fr_mul(w2, z_t)); sub(sc[L.cmA2()], w2); sub(sc[L.T2()], fr_mul(w2, alpha_s));
    add(cs[L.cGu()], fr_mul(w3, z_u)); sub(sc[L.cmB1()], w3); sub(sc[L.U1()], fr_mul(w3, alpha_s));
    add(sc[L.Sp()], fr_mul(w4, z_k)); add(cs[L.cH()], fr_mul(w4, z_u)); sub(sc[L.cmB2()], w4); sub(sc[L.U2()], fr_mul(w4, alpha_s));
  }

  // ---- curdleproofs.py:204-246 + same_msm.py:194-227
  uint8_t Z48[48];
  memset(Z48, 0, 48);
  Z48[0] = 0xC0;
  tr.point("same_msm_step1", Ap48);
  tr.point("same_msm_step1", P(L.T2()));
  tr.point("same_msm_step1", P(L.U2()));
  for (size_t i = 0; i < ell; ++i) tr.point("same_msm_step1", P(L.T(i)));
  tr.point("same_msm_step1", Z48); tr.point("same_msm_step1", Z48); tr.point("same_msm_step1", crs.H48()); tr.point("same_msm_step1", Z48);
  for (size_t i = 0; i < ell; ++i) tr.point("same_msm_step1", P(L.U(i)));
  tr.point("same_msm_step1", Z48); tr.point("same_msm_step1", Z48); tr.point("same_msm_step1", Z48); tr.point("same_msm_step1", crs.H48());
  tr.point("same_msm_step1", P(L.Ba()));
  tr.point("same_msm_step1", P(L.Bt()));
  tr.point("same_msm_step1", P(L.Bu()));
  const fr alpha_m = tr.challenge("same_msm_alpha");
  std::vector<fr> gm(lg), gm_inv;
  for (size_t j = 0; j < lg; ++j) {
    tr.point("same_msm_loop", P(L.LA(j))); tr.point("same_msm_loop", P(L.LT(j))); tr.point("same_msm_loop", P(L.LU(j)));
    tr.point("same_msm_loop", P(L.RA(j))); tr.point("same_msm_loop", P(L.RT(j))); tr.point("same_msm_loop", P(L.RU(j)));
    gm[j] = tr.challenge("same_msm_gamma");
  }
  gm_inv = gm;
  fr_batch_inv(gm_inv.data(), lg);
  if (out_rowin) {
    const fr head[8] = {alpha_p, beta_p, alpha_g, beta_g, alpha_i, beta_i, alpha_s, alpha_m};
    const fr fields[6] = {c_fin, d_fin, z_k, z_t, z_u, x_fin};
    write_rowin(out_rowin, ell, lg, head, gam.data(), gm.data(), a.data(), beta_inv, inner_prod, gam_inv.data(), gm_inv.data(), fields, rho);
    return 0;
  }
  std::vector<fr> sm;
  fold_scalars(gm, sm);
  {
    const fr w4 = rho[3], w5 = rho[4], w6 = rho[5];
    for (size_t j = 0; j < lg; ++j) {
      add(sc[L.LA(j)], fr_mul(w4, gm[j])); add(sc[L.RA(j)], fr_mul(w4, gm_inv[j]));
      add(sc[L.LT(j)], fr_mul(w5, gm[j])); add(sc[L.RT(j)], fr_mul(w5, gm_inv[j]));
      add(sc[L.LU(j)], fr_mul(w6, gm[j])); add(sc[L.RU(j)], fr_mul(w6, gm_inv[j]));
    }
    add(sc[L.Ba()], w4); add(sc[L.Bt()], w5); add(sc[L.Bu()], w6);
    // alpha A' with A' = A + T_1 + U_1;  alpha Z_t, alpha Z_u with Z_t = cm_T.T_2, Z_u = cm_U.T_2
    const fr w4a = fr_mul(w4, alpha_m);
    add(sc[L.A()], w4a); add(sc[L.T1()], w4a); add(sc[L.U1()], w4a);
    add(sc[L.T2()], fr_mul(w5, alpha_m));
    add(sc[L.U2()], fr_mul(w6, alpha_m));
    const fr x4 = fr_mul(w4, x_fin), x5 = fr_mul(w5, x_fin), x6 = fr_mul(w6, x_fin);
    for (size_t i = 0; i < ell; ++i) {
      sub(cs[L.cG(i)], fr_mul(x4, sm[i]));
      sub(sc[L.T(i)], fr_mul(x5, sm[i]));
      sub(sc[L.U(i)], fr_mul(x6, sm[i]));
    }
    // blinder slots: G'' = ... | vec_H[0] vec_H[1] G_t G_u;  T' = ... | Z Z H Z;  U' = ... | Z Z Z H   (curdleproofs.py:206-224)
    sub(cs[L.cHv(0)], fr_mul(x4, sm[ell])); sub(cs[L.cHv(1)], fr_mul(x4, sm[ell + 1]));
    sub(cs[L.cGt()], fr_mul(x4, sm[ell + 2])); sub(cs[L.cGu()], fr_mul(x4, sm[ell + 3]));
    sub(cs[L.cH()], fr_mul(x5, sm[ell + 2]));
    sub(cs[L.cH()], fr_mul(x6, sm[ell + 3]));
  }

  // ---- curdleproofs.py:239-244:  R = <a, vec_R>,  S = <a, vec_S>
  {
    const fr w7 = rho[6], w8 = rho[7];
    add(sc[L.Rp()], w7); add(sc[L.Sp()], w8);
    for (size_t i = 0; i < ell; ++i) { sub(sc[L.R(i)], fr_mul(w7, a[i])); sub(sc[L.S(i)], fr_mul(w8, a[i])); }
  }

  for (size_t i = 0; i < L.count(); ++i) fr_to_le32(sc[i], out_scalars + 32 * i);
  for (size_t i = 0; i < L.ncrs(); ++i) fr_to_le32(cs[i], out_crs_scalars + 32 * i);
  if (out_challenges) {
    uint8_t* o = out_challenges;
    const fr head[8] = {alpha_p, beta_p, alpha_g, beta_g, alpha_i, beta_i, alpha_s, alpha_m};
    for (const fr& c : head) { fr_to_le32(c, o); o += 32; }
    for (size_t j = 0; j < lg; ++j) { fr_to_le32(gam[j], o); o += 32; }
    for (size_t j = 0; j < lg; ++j) { fr_to_le32(gm[j], o); o += 32; }
    for (size_t i = 0; i < ell; ++i) { fr_to_le32(a[i], o); o += 32; }
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// The same front-end for up to cg1m::G proofs at once: identical arithmetic per proof, but the transcripts advance in
// step so that their Keccak permutations run eight at a time (csrc/merlin_group.h).  Output bytes are identical to
// prepare_one's (tests/test_shuffle_verifier.py::test_grouped_front_end_matches_single).
struct ProofWork {
  int status = 0;
  uint8_t* pts = nullptr;
  fr r_p, c_fin, d_fin, z_k, z_t, z_u, x_fin, rho[12];
  Aff aA, aT1, aU1, aB;
  fr alpha_p, beta_p, gprod, alpha_g, beta_g, beta_inv, inner_prod, alpha_i, beta_i, alpha_s, alpha_m;
  std::vector<fr> a, u, gam, gam_inv, s, s_inv, gm, gm_inv, sm;
  uint8_t D48[48], Ap48[48];
};

#ifdef CG1_FE_PROFILE
#include <chrono>
std::atomic<long long> g_prof[6];                 // ns: parse+decode, transcript, fr (between), ec, rows, total
struct ProfTimer {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  long long lap() { auto n = std::chrono::steady_clock::now(); long long d = std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count(); t = n; return d; }
};
#define PROF_LAP(i) g_prof[i] += prof.lap()
#else
#define PROF_LAP(i)
#endif

void prepare_group(const Crs& crs, int cnt, const uint8_t* const* inst, const uint8_t* const* proof, const uint8_t* const* weights,
                   const uint8_t* const* decoded, uint8_t* const* out_points, uint8_t* const* out_scalars,
                   uint8_t* const* out_crs_scalars, uint8_t* const* out_challenges, int32_t* const* status_out,
                   uint8_t* const* out_rowin = nullptr) {
  const size_t ell = crs.ell, lg = crs.lg, n = ell + NB;
  const Layout L(ell, lg);
  ProofWork w[cg1m::G];
  static const uint8_t zero48[48] = {0};
#ifdef CG1_FE_PROFILE
  ProfTimer prof;
#endif

  // ---- parse (as prepare_one); a proof rejected here keeps running on harmless values and is zeroed at the end
  for (int k = 0; k < cnt; ++k) {
    ProofWork& q = w[k];
    q.pts = out_points[k];
    memcpy(q.pts, inst[k], 4 * ell * 48);
    const uint8_t* p = proof[k];
    uint8_t* o = q.pts + L.base() * 48;
    auto pts = [&](size_t m) { memcpy(o, p, 48 * m); o += 48 * m; p += 48 * m; };
    bool ok = true;
    auto sc = [&](fr& dst) { if (!fr_from_le32(p, dst)) { ok = false; dst = fr_zero(); } p += 32; };
    pts(10); sc(q.r_p);
    pts(2 + 4 * lg); sc(q.c_fin); sc(q.d_fin);
    pts(4); sc(q.z_k); sc(q.z_t); sc(q.z_u);
    pts(3 + 6 * lg); sc(q.x_fin);
    canonicalize_infinities(q.pts, L.count());
    if (!ok) q.status = CG1_SHUFFLE_BAD_SCALAR;
    else if (q.pts[L.T(0) * 48] & 0x40) q.status = CG1_SHUFFLE_T0_INFINITY;
    for (int j = 0; j < 12; ++j)
      if (!fr_from_le32(weights[k] + 32 * j, q.rho[j])) { q.rho[j] = fr_zero(); if (!q.status) q.status = CG1_SHUFFLE_BAD_WEIGHT; }
    auto dec = [&](size_t idx) { return decoded && decoded[k] ? decoded[k] + 96 * (idx - L.A()) : nullptr; };
    auto pt = [&](size_t idx, Aff& dst) {
      if (!decode_point(q.pts + idx * 48, dec(idx), dst)) {
        dst.inf = true; dst.x = cg1h::fe_zero(); dst.y = cg1h::fe_zero();
        if (!q.status) q.status = CG1_SHUFFLE_BAD_POINT;
      }
    };
    if (q.status == 0 || q.status == CG1_SHUFFLE_BAD_WEIGHT) { pt(L.A(), q.aA); pt(L.T1(), q.aT1); pt(L.U1(), q.aU1); pt(L.B(), q.aB); }
    else { q.aA.inf = q.aT1.inf = q.aU1.inf = q.aB.inf = true; }
  }

  PROF_LAP(0);
  cg1m::Group tr;
  tr.init("curdleproofs", cnt);
  const uint8_t* ptrs[cg1m::G];
  uint8_t tmp[cg1m::G][32];
  auto point = [&](const char* label, size_t idx) {            // own wire point idx of every proof
    for (int k = 0; k < cnt; ++k) ptrs[k] = w[k].pts + idx * 48;
    tr.append(label, ptrs, 48);
  };
  auto scalar = [&](const char* label, fr ProofWork::*m) {
    for (int k = 0; k < cnt; ++k) { fr_to_le32(w[k].*m, tmp[k]); ptrs[k] = tmp[k]; }
    tr.append(label, ptrs, 32);
  };
  auto challenge = [&](const char* label, fr ProofWork::*m) {
    tr.challenge_scalar(label, tmp);
    for (int k = 0; k < cnt; ++k) fr_from_le32(tmp[k], w[k].*m);
  };

  // ---- curdleproofs.py:176-180
  for (size_t i = 0; i < 4 * ell; ++i) point("curdleproofs_step1", i);
  point("curdleproofs_step1", L.M());
  for (int k = 0; k < cnt; ++k) w[k].a.resize(ell);
  for (size_t i = 0; i < ell; ++i) {
    tr.challenge_scalar("curdleproofs_vec_a", tmp);
    for (int k = 0; k < cnt; ++k) fr_from_le32(tmp[k], w[k].a[i]);
  }
  // ---- same_perm.py:91-98
  point("same_perm_step1", L.A());
  point("same_perm_step1", L.M());
  for (size_t i = 0; i < ell; ++i) {
    for (int k = 0; k < cnt; ++k) { fr_to_le32(w[k].a[i], tmp[k]); ptrs[k] = tmp[k]; }
    tr.append("same_perm_step1", ptrs, 32);
  }
  challenge("same_perm_alpha", &ProofWork::alpha_p);
  challenge("same_perm_beta", &ProofWork::beta_p);
  PROF_LAP(1);
  for (int k = 0; k < cnt; ++k) {
    ProofWork& q = w[k];
    q.gprod = fr_one();
    fr t = q.beta_p;
    for (size_t i = 0; i < ell; ++i) { q.gprod = fr_mul(q.gprod, fr_add(q.a[i], t)); t = fr_add(t, q.alpha_p); }
  }
  PROF_LAP(2);
  // ---- grand_prod.py:175-199
  point("gprod_step1", L.B());
  scalar("gprod_step1", &ProofWork::gprod);
  challenge("gprod_alpha", &ProofWork::alpha_g);
  point("gprod_step2", L.C());
  scalar("gprod_step2", &ProofWork::r_p);
  challenge("gprod_beta", &ProofWork::beta_g);
  PROF_LAP(1);
  {
    // the group's field inversions share ONE inversion each (Montgomery's trick): beta^-1 of every proof here, the
    // Z coordinates of every proof's D and A' below, all challenge inverses after the transcript
    fr binv[cg1m::G];
    for (int k = 0; k < cnt; ++k) binv[k] = w[k].beta_g;
    fr_batch_inv(binv, (size_t)cnt);
    jac pts2[2 * cg1m::G];
    for (int k = 0; k < cnt; ++k) {
      ProofWork& q = w[k];
      q.beta_inv = binv[k];
      q.u.resize(n);
      fr pw = q.beta_inv;
      for (size_t i = 0; i < ell; ++i) { q.u[i] = pw; pw = fr_mul(pw, q.beta_inv); }
      for (size_t i = ell; i < n; ++i) q.u[i] = pw;
      jac& d = pts2[2 * k];
      jac& ap = pts2[2 * k + 1];
      d = cg1h::jac_identity();
      ap = cg1h::jac_identity();
      crs.g_sum.mul_into(d, fr_neg(q.beta_inv));
      crs.h_sum.mul_into(d, q.alpha_g);
      madd_aff(d, q.aB);
      madd_aff(ap, q.aA); madd_aff(ap, q.aT1); madd_aff(ap, q.aU1);
      const fr beta_ell = fr_pow_u64(q.beta_g, ell);
      q.inner_prod = fr_sub(fr_add(fr_mul(q.r_p, fr_mul(beta_ell, q.beta_g)), fr_mul(q.gprod, beta_ell)), fr_one());
    }
    cg1h::fe xs[2 * cg1m::G], ys[2 * cg1m::G];
    uint8_t inf[2 * cg1m::G];
    cg1h::jac_batch_to_affine(pts2, 2 * (size_t)cnt, xs, ys, inf);
    for (int k = 0; k < cnt; ++k) {
      cg1h::g1_compress_affine(xs[2 * k], ys[2 * k], inf[2 * k] != 0, w[k].D48);
      cg1h::g1_compress_affine(xs[2 * k + 1], ys[2 * k + 1], inf[2 * k + 1] != 0, w[k].Ap48);
    }
  }
  PROF_LAP(3);
  // ---- ipa.py:204-212, 170-176
  point("ipa_step1", L.C());
  for (int k = 0; k < cnt; ++k) ptrs[k] = w[k].D48;
  tr.append("ipa_step1", ptrs, 48);
  scalar("ipa_step1", &ProofWork::inner_prod);
  point("ipa_step1", L.Bc());
  point("ipa_step1", L.Bd());
  challenge("ipa_alpha", &ProofWork::alpha_i);
  challenge("ipa_beta", &ProofWork::beta_i);
  for (int k = 0; k < cnt; ++k) { w[k].gam.resize(lg); w[k].gm.resize(lg); }
  for (size_t j = 0; j < lg; ++j) {
    point("ipa_loop", L.LC(j)); point("ipa_loop", L.LD(j)); point("ipa_loop", L.RC(j)); point("ipa_loop", L.RD(j));
    tr.challenge_scalar("ipa_gamma", tmp);
    for (int k = 0; k < cnt; ++k) fr_from_le32(tmp[k], w[k].gam[j]);
  }
  // ---- same_scalar.py:82-99
  {
    const size_t order[10] = {L.Rp(), L.Sp(), L.T1(), L.T2(), L.U1(), L.U2(), L.cmA1(), L.cmA2(), L.cmB1(), L.cmB2()};
    for (size_t i = 0; i < 10; ++i) point("sameexp_points", order[i]);
  }
  challenge("same_scalar_alpha", &ProofWork::alpha_s);
  // ---- curdleproofs.py:204-224 + same_msm.py:194-204, 164-172
  uint8_t Z48[48];
  memset(Z48, 0, 48);
  Z48[0] = 0xC0;
  for (int k = 0; k < cnt; ++k) ptrs[k] = w[k].Ap48;
  tr.append("same_msm_step1", ptrs, 48);
  point("same_msm_step1", L.T2());
  point("same_msm_step1", L.U2());
  for (size_t i = 0; i < ell; ++i) point("same_msm_step1", L.T(i));
  tr.append_same("same_msm_step1", Z48, 48); tr.append_same("same_msm_step1", Z48, 48);
  tr.append_same("same_msm_step1", crs.H48(), 48); tr.append_same("same_msm_step1", Z48, 48);
  for (size_t i = 0; i < ell; ++i) point("same_msm_step1", L.U(i));
  tr.append_same("same_msm_step1", Z48, 48); tr.append_same("same_msm_step1", Z48, 48);
  tr.append_same("same_msm_step1", Z48, 48); tr.append_same("same_msm_step1", crs.H48(), 48);
  point("same_msm_step1", L.Ba()); point("same_msm_step1", L.Bt()); point("same_msm_step1", L.Bu());
  challenge("same_msm_alpha", &ProofWork::alpha_m);
  for (size_t j = 0; j < lg; ++j) {
    point("same_msm_loop", L.LA(j)); point("same_msm_loop", L.LT(j)); point("same_msm_loop", L.LU(j));
    point("same_msm_loop", L.RA(j)); point("same_msm_loop", L.RT(j)); point("same_msm_loop", L.RU(j));
    tr.challenge_scalar("same_msm_gamma", tmp);
    for (int k = 0; k < cnt; ++k) fr_from_le32(tmp[k], w[k].gm[j]);
  }
  (void)zero48;
  PROF_LAP(1);
  {
    std::vector<fr> all(2 * lg * (size_t)cnt);
    for (int k = 0; k < cnt; ++k)
      for (size_t j = 0; j < lg; ++j) { all[(2 * k) * lg + j] = w[k].gam[j]; all[(2 * k + 1) * lg + j] = w[k].gm[j]; }
    fr_batch_inv(all.data(), all.size());
    for (int k = 0; k < cnt; ++k) {
      w[k].gam_inv.assign(all.begin() + (2 * k) * lg, all.begin() + (2 * k + 1) * lg);
      w[k].gm_inv.assign(all.begin() + (2 * k + 1) * lg, all.begin() + (2 * k + 2) * lg);
    }
  }

  // ---- the scalar rows (all equations as in prepare_one)
  for (int k = 0; k < cnt; ++k) {
    ProofWork& q = w[k];
    *status_out[k] = q.status;
    if (out_rowin) {                                   // rows are built on the device from this block (k_shuffle_rows)
      if (q.status) { memset(out_rowin[k], 0, rowin_scalars(ell, lg) * 32); continue; }
      const fr head[8] = {q.alpha_p, q.beta_p, q.alpha_g, q.beta_g, q.alpha_i, q.beta_i, q.alpha_s, q.alpha_m};
      const fr fields[6] = {q.c_fin, q.d_fin, q.z_k, q.z_t, q.z_u, q.x_fin};
      write_rowin(out_rowin[k], ell, lg, head, q.gam.data(), q.gm.data(), q.a.data(), q.beta_inv, q.inner_prod, q.gam_inv.data(),
                  q.gm_inv.data(), fields, q.rho);
      continue;
    }
    if (q.status) {
      memset(out_scalars[k], 0, L.count() * 32);
      memset(out_crs_scalars[k], 0, L.ncrs() * 32);
      continue;
    }
    std::vector<fr> sc(L.count(), fr_zero()), cs(L.ncrs(), fr_zero());
    auto add = [](fr& dst, const fr& v) { dst = fr_add(dst, v); };
    auto sub = [](fr& dst, const fr& v) { dst = fr_sub(dst, v); };
    const fr* rho = q.rho;
    // E1
    add(sc[L.B()], rho[0]); sub(sc[L.A()], rho[0]); sub(sc[L.M()], fr_mul(rho[0], q.alpha_p));
    {
      const fr t = fr_mul(rho[0], q.beta_p);
      for (size_t i = 0; i < ell; ++i) sub(cs[L.cG(i)], t);
    }
    fold_scalars(q.gam, q.s);
    fold_scalars(q.gam_inv, q.s_inv);
    {   // E2
      const fr wt = rho[1];
      for (size_t j = 0; j < lg; ++j) { add(sc[L.LC(j)], fr_mul(wt, q.gam[j])); add(sc[L.RC(j)], fr_mul(wt, q.gam_inv[j])); }
      add(sc[L.Bc()], wt);
      add(sc[L.C()], fr_mul(wt, q.alpha_i));
      const fr hcoef = fr_mul(q.beta_i, fr_sub(fr_mul(fr_sqr(q.alpha_i), q.inner_prod), fr_mul(q.c_fin, q.d_fin)));
      add(cs[L.cH()], fr_mul(wt, hcoef));
      const fr wc = fr_mul(wt, q.c_fin);
      for (size_t i = 0; i < n; ++i) sub(cs[i], fr_mul(wc, q.s[i]));
    }
    {   // E3
      const fr wt = rho[2];
      for (size_t j = 0; j < lg; ++j) { add(sc[L.LD(j)], fr_mul(wt, q.gam[j])); add(sc[L.RD(j)], fr_mul(wt, q.gam_inv[j])); }
      add(sc[L.Bd()], wt);
      const fr wa = fr_mul(wt, q.alpha_i);
      add(sc[L.B()], wa);
      sub(cs[L.cGsum()], fr_mul(wa, q.beta_inv));
      add(cs[L.cHsum()], fr_mul(wa, q.alpha_g));
      const fr wd = fr_mul(wt, q.d_fin);
      for (size_t i = 0; i < n; ++i) sub(cs[i], fr_mul(wd, fr_mul(q.s_inv[i], q.u[i])));
    }
    {   // same-scalar
      const fr w1 = rho[8], w2 = rho[9], w3 = rho[10], w4 = rho[11];
      add(cs[L.cGt()], fr_mul(w1, q.z_t)); sub(sc[L.cmA1()], w1); sub(sc[L.T1()], fr_mul(w1, q.alpha_s));
      add(sc[L.Rp()], fr_mul(w2, q.z_k)); add(cs[L.cH()], fr_mul(w2, q.z_t)); sub(sc[L.cmA2()], w2); sub(sc[L.T2()], fr_mul(w2, q.alpha_s));
      add(cs[L.cGu()], fr_mul(w3, q.z_u)); sub(sc[L.cmB1()], w3); sub(sc[L.U1()], fr_mul(w3, q.alpha_s));
      add(sc[L.Sp()], fr_mul(w4, q.z_k)); add(cs[L.cH()], fr_mul(w4, q.z_u)); sub(sc[L.cmB2()], w4); sub(sc[L.U2()], fr_mul(w4, q.alpha_s));
    }
    fold_scalars(q.gm, q.sm);
    {   // E4-E6
      const fr w4 = rho[3], w5 = rho[4], w6 = rho[5];
      for (size_t j = 0; j < lg; ++j) {
        add(sc[L.LA(j)], fr_mul(w4, q.gm[j])); add(sc[L.RA(j)], fr_mul(w4, q.gm_inv[j]));
        add(sc[L.LT(j)], fr_mul(w5, q.gm[j])); add(sc[L.RT(j)], fr_mul(w5, q.gm_inv[j]));
        add(sc[L.LU(j)], fr_mul(w6, q.gm[j])); add(sc[L.RU(j)], fr_mul(w6, q.gm_inv[j]));
      }
      add(sc[L.Ba()], w4); add(sc[L.Bt()], w5); add(sc[L.Bu()], w6);
      const fr w4a = fr_mul(w4, q.alpha_m);
      add(sc[L.A()], w4a); add(sc[L.T1()], w4a); add(sc[L.U1()], w4a);
      add(sc[L.T2()], fr_mul(w5, q.alpha_m));
      add(sc[L.U2()], fr_mul(w6, q.alpha_m));
      const fr x4 = fr_mul(w4, q.x_fin), x5 = fr_mul(w5, q.x_fin), x6 = fr_mul(w6, q.x_fin);
      for (size_t i = 0; i < ell; ++i) {
        sub(cs[L.cG(i)], fr_mul(x4, q.sm[i]));
        sub(sc[L.T(i)], fr_mul(x5, q.sm[i]));
        sub(sc[L.U(i)], fr_mul(x6, q.sm[i]));
      }
      sub(cs[L.cHv(0)], fr_mul(x4, q.sm[ell])); sub(cs[L.cHv(1)], fr_mul(x4, q.sm[ell + 1]));
      sub(cs[L.cGt()], fr_mul(x4, q.sm[ell + 2])); sub(cs[L.cGu()], fr_mul(x4, q.sm[ell + 3]));
      sub(cs[L.cH()], fr_mul(x5, q.sm[ell + 2]));
      sub(cs[L.cH()], fr_mul(x6, q.sm[ell + 3]));
    }
    {   // E7, E8
      const fr w7 = rho[6], w8 = rho[7];
      add(sc[L.Rp()], w7); add(sc[L.Sp()], w8);
      for (size_t i = 0; i < ell; ++i) { sub(sc[L.R(i)], fr_mul(w7, q.a[i])); sub(sc[L.S(i)], fr_mul(w8, q.a[i])); }
    }
    for (size_t i = 0; i < L.count(); ++i) fr_to_le32(sc[i], out_scalars[k] + 32 * i);
    for (size_t i = 0; i < L.ncrs(); ++i) fr_to_le32(cs[i], out_crs_scalars[k] + 32 * i);
    if (out_challenges && out_challenges[k]) {
      uint8_t* o = out_challenges[k];
      const fr head[8] = {q.alpha_p, q.beta_p, q.alpha_g, q.beta_g, q.alpha_i, q.beta_i, q.alpha_s, q.alpha_m};
      for (const fr& c : head) { fr_to_le32(c, o); o += 32; }
      for (size_t j = 0; j < lg; ++j) { fr_to_le32(q.gam[j], o); o += 32; }
      for (size_t j = 0; j < lg; ++j) { fr_to_le32(q.gm[j], o); o += 32; }
      for (size_t i = 0; i < ell; ++i) { fr_to_le32(q.a[i], o); o += 32; }
    }
  }
  PROF_LAP(4);
}

std::atomic<int> g_grouped{1};

}  // namespace
#ifdef CG1_FE_PROFILE
std::atomic<long long> g_tr_prof[4];
extern "C" void cg1_shuffle_profile(double out_ms[6]) { for (int i = 0; i < 6; ++i) out_ms[i] = g_prof[i].exchange(0) * 1e-6; }
extern "C" void cg1_transcript_profile(double out[4]) { for (int i = 0; i < 3; ++i) out[i] = g_tr_prof[i].exchange(0) * 1e-6; out[3] = (double)g_tr_prof[3].exchange(0); }
#endif
namespace {
}

extern "C" {

// 1 (default): proofs are prepared cg1m::G at a time with batched Keccak permutations; 0: one transcript at a time
void cg1_shuffle_set_grouped(int on) { g_grouped.store(on ? 1 : 0); }


cg1_shuffle_crs* cg1_shuffle_crs_create(const uint8_t* crs_bytes, size_t ell, size_t n_blinders) {
  if (n_blinders != NB || ell == 0) return nullptr;
  size_t n = ell + NB, lg = 0;
  while (((size_t)1 << lg) < n) ++lg;
  if (((size_t)1 << lg) != n || lg >= 32) return nullptr;      // crs.py:50-52, ipa.py:160-163
  Crs* c = new Crs;
  c->ell = ell;
  c->lg = lg;
  c->bytes.assign(crs_bytes, crs_bytes + (ell + NB + 5) * 48);
  jac gs, hs;
  for (size_t i = 0; i < ell + NB + 5; ++i) {                  // every CRS encoding must be valid (crs.py:104-113)
    jac t;
    if (cg1h::g1_decompress(crs_bytes + 48 * i, false, t)) { delete c; return nullptr; }
    if (i == ell + NB + 3) gs = t;
    if (i == ell + NB + 4) hs = t;
  }
  c->g_sum.build(gs);
  c->h_sum.build(hs);
  return reinterpret_cast<cg1_shuffle_crs*>(c);
}

void cg1_shuffle_crs_destroy(cg1_shuffle_crs* crs) { delete reinterpret_cast<Crs*>(crs); }

size_t cg1_shuffle_proof_bytes(const cg1_shuffle_crs* crs) { return proof_wire_bytes(reinterpret_cast<const Crs*>(crs)->lg); }
size_t cg1_shuffle_points_per_proof(const cg1_shuffle_crs* crs) {
  const Crs* c = reinterpret_cast<const Crs*>(crs);
  return Layout(c->ell, c->lg).count();
}
size_t cg1_shuffle_crs_points(const cg1_shuffle_crs* crs) {
  const Crs* c = reinterpret_cast<const Crs*>(crs);
  return Layout(c->ell, c->lg).ncrs();
}
size_t cg1_shuffle_challenges_per_proof(const cg1_shuffle_crs* crs) {
  const Crs* c = reinterpret_cast<const Crs*>(crs);
  return 8 + 2 * c->lg + c->ell;
}

// worker threads cg1_shuffle_prepare uses with n_threads = 0 (usable CPUs: affinity mask capped by the cgroup CPU quota;
// CURDLE_G1_THREADS overrides)
size_t cg1_shuffle_default_threads(void) { return Pool::get().size() + 1; }

}  // extern "C"

// out_rowin != NULL: emit the device row builder's input blocks (n x rowin_scalars x 32 B) instead of the rows themselves
static int prepare_dispatch(const cg1_shuffle_crs* crs_, size_t n_proofs, const uint8_t* instances, const uint8_t* proofs,
                            const uint8_t* weights, const uint8_t* decoded96, size_t decoded_stride, uint8_t* out_points48, uint8_t* out_scalars32, uint8_t* out_crs_scalars32,
                            int32_t* status, uint8_t* out_challenges32, int n_threads, uint8_t* out_rowin) {
  if (!crs_ || (n_proofs && (!instances || !proofs || !weights || !out_points48 || !status)))
    return CG1_ERR_ARG;
  if (n_proofs && !out_rowin && (!out_scalars32 || !out_crs_scalars32)) return CG1_ERR_ARG;
  const Crs& crs = *reinterpret_cast<const Crs*>(crs_);
  const Layout L(crs.ell, crs.lg);
  const size_t inst_b = 4 * crs.ell * 48, proof_b = proof_wire_bytes(crs.lg), nch = 8 + 2 * crs.lg + crs.ell;
  const size_t rin_b = rowin_scalars(crs.ell, crs.lg) * 32;
  std::atomic<size_t> next{0};
  const bool grouped = g_grouped.load() != 0;
  const size_t per_item = grouped ? (size_t)cg1m::G : 1, n_items = (n_proofs + per_item - 1) / per_item;
  auto work = [&]() {
    for (;;) {
      const size_t item = next.fetch_add(1);
      if (item >= n_items) return;
      if (grouped) {
        const size_t lo = item * per_item, cnt = std::min(per_item, n_proofs - lo);
        const uint8_t *in[cg1m::G], *pr[cg1m::G], *we[cg1m::G], *de[cg1m::G];
        uint8_t *pts[cg1m::G], *scs[cg1m::G], *ccs[cg1m::G], *chs[cg1m::G], *rin[cg1m::G];
        int32_t* sts[cg1m::G];
        for (size_t k = 0; k < cnt; ++k) {
          const size_t i = lo + k;
          in[k] = instances + i * inst_b; pr[k] = proofs + i * proof_b; we[k] = weights + i * 12 * 32;
          de[k] = decoded96 ? decoded96 + i * decoded_stride : nullptr;
          pts[k] = out_points48 + i * L.count() * 48;
          scs[k] = out_rowin ? nullptr : out_scalars32 + i * L.count() * 32;
          ccs[k] = out_rowin ? nullptr : out_crs_scalars32 + i * L.ncrs() * 32;
          rin[k] = out_rowin ? out_rowin + i * rin_b : nullptr;
          chs[k] = out_challenges32 ? out_challenges32 + i * nch * 32 : nullptr;
          sts[k] = status + i;
        }
        prepare_group(crs, (int)cnt, in, pr, we, de, pts, scs, ccs, chs, sts, out_rowin ? rin : nullptr);
        continue;
      }
      const size_t i = item;
      uint8_t* pts = out_points48 + i * L.count() * 48;
      uint8_t* scs = out_rowin ? nullptr : out_scalars32 + i * L.count() * 32;
      uint8_t* ccs = out_rowin ? nullptr : out_crs_scalars32 + i * L.ncrs() * 32;
      int rc = prepare_one(crs, instances + i * inst_b, proofs + i * proof_b, weights + i * 12 * 32,
                           decoded96 ? decoded96 + i * decoded_stride : nullptr, pts, scs, ccs,
                           out_challenges32 ? out_challenges32 + i * nch * 32 : nullptr, nullptr, out_rowin ? out_rowin + i * rin_b : nullptr);
      if (rc) {                               // a rejected proof contributes nothing to a merged check
        if (out_rowin) memset(out_rowin + i * rin_b, 0, rin_b);
        else { memset(scs, 0, L.count() * 32); memset(ccs, 0, L.ncrs() * 32); }
      }
      status[i] = rc;
    }
  };
  if (n_threads == 1 || n_items <= 1) {
    work();
  } else {
    Pool& pool = Pool::get();
    size_t nt = n_threads > 0 ? (size_t)n_threads : pool.size() + 1;
    if (nt > n_items) nt = n_items;
    pool.run(work, nt);
  }
  return CG1_OK;
}

extern "C" {

int cg1_shuffle_prepare(const cg1_shuffle_crs* crs_, size_t n_proofs, const uint8_t* instances, const uint8_t* proofs,
                        const uint8_t* weights, const uint8_t* decoded96, size_t decoded_stride, uint8_t* out_points48, uint8_t* out_scalars32, uint8_t* out_crs_scalars32,
                        int32_t* status, uint8_t* out_challenges32, int n_threads) {
  return prepare_dispatch(crs_, n_proofs, instances, proofs, weights, decoded96, decoded_stride, out_points48, out_scalars32, out_crs_scalars32,
                          status, out_challenges32, n_threads, nullptr);
}

// The same front-end, but instead of the scalar rows it emits, per proof, the input block of the DEVICE row builder
// (cg1_shuffle_rows_device): cg1_shuffle_rowin_scalars(crs) 32-byte scalars (challenges, their inverses, inner_prod,
// beta^-1, the proof's Fr fields, the weights).  A rejected proof gets a zero block and its code in status[].
size_t cg1_shuffle_rowin_scalars(const cg1_shuffle_crs* crs) {
  const Crs* c = reinterpret_cast<const Crs*>(crs);
  return rowin_scalars(c->ell, c->lg);
}
int cg1_shuffle_prepare_inputs(const cg1_shuffle_crs* crs_, size_t n_proofs, const uint8_t* instances, const uint8_t* proofs,
                               const uint8_t* weights, const uint8_t* decoded96, size_t decoded_stride, uint8_t* out_points48,
                               uint8_t* out_rowin32, int32_t* status, int n_threads) {
  if (n_proofs && !out_rowin32) return CG1_ERR_ARG;
  return prepare_dispatch(crs_, n_proofs, instances, proofs, weights, decoded96, decoded_stride, out_points48, nullptr, nullptr,
                          status, nullptr, n_threads, out_rowin32);
}

// ---- Whisk tracker-opening proofs (opening.py:21-79; IsValidWhiskOpeningProof, whisk_interface.py:147-169), batched the
// same way: proof of knowledge of k with k_r_G = k * r_G and k_G = k * G.  The reference checks
//     s*G + c*k_G == A   and   s*r_G + c*k_r_G == B        (opening.py:74-77)
// with c drawn from the transcript; here both equalities of every proof, weighted by two random rho, become scalars over
// the proof's own points [k_G, k_r_G, r_G, A, B] and over the generator G (whose scalars add up across proofs).
//   trackers: n x (r_G | k_r_G) encodings; k_commitments: n x 48; proofs: n x 128 (A | B | s); weights: n x 2 x 32.
int cg1_opening_prepare(size_t n, const uint8_t* trackers, const uint8_t* k_commitments, const uint8_t* proofs, const uint8_t* weights,
                        uint8_t* out_points48, uint8_t* out_scalars32, uint8_t* out_g_scalars32, int32_t* status) {
  if (n && (!trackers || !k_commitments || !proofs || !weights || !out_points48 || !out_scalars32 || !out_g_scalars32 || !status))
    return CG1_ERR_ARG;
  uint8_t G48[48];
  cg1h::g1_compress(cg1h::jac_generator(), G48);
  auto one = [&](size_t i) {
    const uint8_t *rG = trackers + 96 * i, *krG = rG + 48, *kG = k_commitments + 48 * i, *A = proofs + 128 * i, *B = A + 48;
    uint8_t* pts = out_points48 + 5 * 48 * i;
    memcpy(pts, kG, 48); memcpy(pts + 48, krG, 48); memcpy(pts + 96, rG, 48); memcpy(pts + 144, A, 48); memcpy(pts + 192, B, 48);
    canonicalize_infinities(pts, 5);
    uint8_t* sc = out_scalars32 + 5 * 32 * i;
    uint8_t* gs = out_g_scalars32 + 32 * i;
    memset(sc, 0, 5 * 32);
    memset(gs, 0, 32);
    fr s_, r1, r2;
    if (!fr_from_le32(A + 96, s_)) { status[i] = CG1_SHUFFLE_BAD_SCALAR; return; }
    if (!fr_from_le32(weights + 64 * i, r1) || !fr_from_le32(weights + 64 * i + 32, r2)) { status[i] = CG1_SHUFFLE_BAD_WEIGHT; return; }
    Transcript tr("whisk_opening_proof");
    const uint8_t* order[6] = {pts, G48, pts + 48, pts + 96, pts + 144, pts + 192};          // k_G G k_r_G r_G A B, as re-serialised
    for (const uint8_t* p : order) tr.point("tracker_opening_proof", p);
    const fr c = tr.challenge("tracker_opening_proof_challenge");
    fr_to_le32(fr_mul(r1, c), sc);                 // k_G
    fr_to_le32(fr_mul(r2, c), sc + 32);            // k_r_G
    fr_to_le32(fr_mul(r2, s_), sc + 64);           // r_G
    fr_to_le32(fr_neg(r1), sc + 96);               // A
    fr_to_le32(fr_neg(r2), sc + 128);              // B
    fr_to_le32(fr_mul(r1, s_), gs);                // G
    status[i] = 0;
  };
  if (n < 256) {
    for (size_t i = 0; i < n; ++i) one(i);
  } else {                                                    // slices of 64 proofs over the worker pool
    std::atomic<size_t> next{0};
    const size_t items = (n + 63) / 64;
    std::function<void()> work = [&]() {
      for (;;) {
        const size_t it = next.fetch_add(1);
        if (it >= items) return;
        for (size_t i = it * 64; i < std::min(n, it * 64 + 64); ++i) one(i);
      }
    };
    Pool& pool = Pool::get();
    pool.run(work, std::min(items, pool.size() + 1));
  }
  return CG1_OK;
}

// out[i] = addend[i] + scalars[i % nscalars] * bases[i % nbase] on the HOST's worker pool: the same contract as the device kernel behind
// cg1_batch_mul_add (affine96 standard-form records in and out, zeros = identity, inputs not checked against the curve), for the calls
// where a launch costs more than the arithmetic: a fold / map of a few hundred points is 255 dependent doublings on the GPU whatever
// its size (~2.2 ms), and n scalar multiplications at ~80 us over the pool's threads here.  cg1_batch_mul_add (msm_gpu.hip) chooses.
int cg1_batch_mul_add_pool(const uint8_t* bases, size_t nbase, const uint8_t* scalars, size_t nscalars, const uint8_t* addend, uint8_t* out,
                           size_t n, int n_threads) {
  if (n == 0) return CG1_OK;
  if (!bases || !scalars || !out || nbase == 0 || nscalars == 0) return CG1_ERR_ARG;
  using cg1h::fe; using cg1h::jac;
  std::atomic<int> bad{0};
  std::atomic<size_t> next{0};
  const size_t slice = 16, items = (n + slice - 1) / slice;
  auto load = [&](const uint8_t* rec, jac& o) {
    bool zero = true;
    for (int k = 0; k < 96 && zero; ++k) zero = rec[k] == 0;
    if (zero) { o = cg1h::jac_identity(); return true; }
    fe x, y;
    if (!cg1h::fe_from_le48(rec, x) || !cg1h::fe_from_le48(rec + 48, y)) return false;
    o = cg1h::jac_from_affine(x, y);
    return true;
  };
  std::function<void()> work = [&]() {
    jac acc[slice];
    fe xs[slice], ys[slice];
    uint8_t inf[slice];
    for (;;) {
      const size_t it = next.fetch_add(1);
      if (it >= items) return;
      const size_t lo = it * slice, cnt = std::min(slice, n - lo);
      for (size_t k = 0; k < cnt; ++k) {
        const size_t i = lo + k;
        jac b, a = cg1h::jac_identity();
        if (!load(bases + 96 * (i % nbase), b) || (addend && !load(addend + 96 * i, a))) { bad.store(1); acc[k] = cg1h::jac_identity(); continue; }
        acc[k] = cg1h::jac_add(a, cg1h::jac_mul(b, scalars + 32 * (i % nscalars)));
      }
      cg1h::jac_batch_to_affine(acc, cnt, xs, ys, inf);                  // one inversion per slice
      for (size_t k = 0; k < cnt; ++k) {
        uint8_t* o = out + 96 * (lo + k);
        if (inf[k]) { memset(o, 0, 96); continue; }
        cg1h::fe_to_le48(xs[k], o);
        cg1h::fe_to_le48(ys[k], o + 48);
      }
    }
  };
  Pool& pool = Pool::get();
  size_t nt = n_threads > 0 ? (size_t)n_threads : pool.size() + 1;
  nt = std::min(nt, items);
  if (nt <= 1) work(); else pool.run(work, nt);
  return bad.load() ? CG1_ERR_ENCODING : CG1_OK;
}

// rho1 | rho2 of proof i from a per-batch seed: the first 32 bytes of SHAKE256(seed || le64(i)), 16 bytes each (the device's
// weights_from_seed, csrc/kernels_opening.h, computes the very same)
int cg1_opening_weights_from_seed(const uint8_t seed32[32], size_t first, size_t n, uint8_t* out_weights64) {
  if (!seed32 || (n && !out_weights64)) return CG1_ERR_ARG;
  for (size_t k = 0; k < n; ++k) {
    uint8_t st[200];
    memset(st, 0, sizeof st);
    memcpy(st, seed32, 32);
    const uint64_t i = (uint64_t)(first + k);
    for (int b = 0; b < 8; ++b) st[32 + b] = (uint8_t)(i >> (8 * b));
    st[40] ^= 0x1F;
    st[135] ^= 0x80;
    cg1_keccak_f1600(st);
    uint8_t* o = out_weights64 + 64 * k;
    memset(o, 0, 64);
    memcpy(o, st, 16);
    memcpy(o + 32, st + 16, 16);
  }
  return CG1_OK;
}

// The equalities the reference asserts directly, evaluated EXACTLY (no random weights) with host group arithmetic, for
// proofs that carry a point outside G1: random weights are only sound inside the prime-order subgroup (E(Fp) has points of
// order 3, 11, ...), and the reference's own verdict on such a proof is the exact one.
int cg1_shuffle_exact_same_scalar(const cg1_shuffle_crs* crs_, const uint8_t* instance, const uint8_t* proof, int* ok) {
  if (!crs_ || !instance || !proof || !ok) return CG1_ERR_ARG;
  *ok = 0;
  const Crs& crs = *reinterpret_cast<const Crs*>(crs_);
  const Layout L(crs.ell, crs.lg);
  std::vector<uint8_t> pts(L.count() * 48), sc(L.count() * 32), cs(L.ncrs() * 32);
  uint8_t w[12 * 32];
  memset(w, 0, sizeof w);
  for (int k = 0; k < 12; ++k) w[32 * k] = 1;                       // any valid weights: only alpha is wanted
  fr alpha;
  if (prepare_one(crs, instance, proof, w, nullptr, pts.data(), sc.data(), cs.data(), nullptr, &alpha) != 0) return CG1_OK;
  fr z_k, z_t, z_u;
  const uint8_t* zs = proof + 48 * (16 + 4 * crs.lg) + 32 * 3;      // M..C (10) r_p | B_c B_d L/R_C L/R_D (2+4lg) c d | cm_A cm_B (4) | z_k z_t z_u
  if (!fr_from_le32(zs, z_k) || !fr_from_le32(zs + 32, z_t) || !fr_from_le32(zs + 64, z_u)) return CG1_OK;
  auto dec = [&](const uint8_t* e, jac& out) { return cg1h::g1_decompress(e, false, out) == 0; };
  auto P = [&](size_t idx) { return pts.data() + idx * 48; };
  jac R, S, T1, T2, U1, U2, A1, A2, B1, B2, Gt, Gu, H;
  if (!dec(P(L.Rp()), R) || !dec(P(L.Sp()), S) || !dec(P(L.T1()), T1) || !dec(P(L.T2()), T2) || !dec(P(L.U1()), U1) ||
      !dec(P(L.U2()), U2) || !dec(P(L.cmA1()), A1) || !dec(P(L.cmA2()), A2) || !dec(P(L.cmB1()), B1) || !dec(P(L.cmB2()), B2) ||
      !dec(crs.bytes.data() + L.cGt() * 48, Gt) || !dec(crs.bytes.data() + L.cGu() * 48, Gu) || !dec(crs.H48(), H))
    return CG1_OK;
  uint8_t a32[32], zk32[32], zt32[32], zu32[32];
  fr_to_le32(alpha, a32); fr_to_le32(z_k, zk32); fr_to_le32(z_t, zt32); fr_to_le32(z_u, zu32);
  using cg1h::jac_add; using cg1h::jac_mul; using cg1h::jac_eq;
  // GroupCommitment.new(G, H, T, r) = (G r, T + H r)  (commitment.py:22-30); expected == cm + cm' alpha (same_scalar.py:101-108)
  const bool e1 = jac_eq(jac_mul(Gt, zt32), jac_add(A1, jac_mul(T1, a32)));
  const bool e2 = jac_eq(jac_add(jac_mul(R, zk32), jac_mul(H, zt32)), jac_add(A2, jac_mul(T2, a32)));
  const bool e3 = jac_eq(jac_mul(Gu, zu32), jac_add(B1, jac_mul(U1, a32)));
  const bool e4 = jac_eq(jac_add(jac_mul(S, zk32), jac_mul(H, zu32)), jac_add(B2, jac_mul(U2, a32)));
  *ok = (e1 && e2 && e3 && e4) ? 1 : 0;
  return CG1_OK;
}

// status: 0 accepted, CG1_SHUFFLE_BAD_SCALAR (s >= r), CG1_SHUFFLE_BAD_POINT (a point does not decode), 6 (an equality fails) -- the codes
// the batch path reports (OpeningBatchVerifier.last_status), in its order of checks
int cg1_opening_exact_status(const uint8_t* tracker96, const uint8_t* k_commitment48, const uint8_t* proof128, int* status);
int cg1_opening_exact(const uint8_t* tracker96, const uint8_t* k_commitment48, const uint8_t* proof128, int* ok) {
  if (!ok) return CG1_ERR_ARG;
  int st = 0;
  const int rc = cg1_opening_exact_status(tracker96, k_commitment48, proof128, &st);
  *ok = (rc == CG1_OK && st == 0) ? 1 : 0;
  return rc;
}
int cg1_opening_exact_status(const uint8_t* tracker96, const uint8_t* k_commitment48, const uint8_t* proof128, int* status) {
  if (!tracker96 || !k_commitment48 || !proof128 || !status) return CG1_ERR_ARG;
  int st_ok = 0;
  int* ok = &st_ok;
  *status = 6;
  const uint8_t *rG = tracker96, *krG = tracker96 + 48, *kG = k_commitment48, *A = proof128, *B = proof128 + 48;
  fr s_;
  if (!fr_from_le32(proof128 + 96, s_)) { *status = CG1_SHUFFLE_BAD_SCALAR; return CG1_OK; }
  jac jrG, jkrG, jkG, jA, jB;
  if (cg1h::g1_decompress(rG, false, jrG) || cg1h::g1_decompress(krG, false, jkrG) || cg1h::g1_decompress(kG, false, jkG) ||
      cg1h::g1_decompress(A, false, jA) || cg1h::g1_decompress(B, false, jB)) {
    *status = CG1_SHUFFLE_BAD_POINT;
    return CG1_OK;
  }
  uint8_t G48[48], own[5 * 48];
  cg1h::g1_compress(cg1h::jac_generator(), G48);
  memcpy(own, kG, 48); memcpy(own + 48, krG, 48); memcpy(own + 96, rG, 48); memcpy(own + 144, A, 48); memcpy(own + 192, B, 48);
  canonicalize_infinities(own, 5);
  Transcript tr("whisk_opening_proof");
  const uint8_t* order[6] = {own, G48, own + 48, own + 96, own + 144, own + 192};
  for (const uint8_t* p : order) tr.point("tracker_opening_proof", p);
  const fr c = tr.challenge("tracker_opening_proof_challenge");
  uint8_t c32[32], s32[32];
  fr_to_le32(c, c32); fr_to_le32(s_, s32);
  using cg1h::jac_add; using cg1h::jac_mul; using cg1h::jac_eq;
  const bool e1 = jac_eq(jac_add(jac_mul(cg1h::jac_generator(), s32), jac_mul(jkG, c32)), jA);      // opening.py:73
  const bool e2 = jac_eq(jac_add(jac_mul(jrG, s32), jac_mul(jkrG, c32)), jB);                       // opening.py:74
  *ok = (e1 && e2) ? 1 : 0;
  *status = *ok ? 0 : 6;
  return CG1_OK;
}

// Only the first step of cg1_shuffle_prepare: the proofs' own points in layout order (what the GPU decompresses).
int cg1_shuffle_gather_points(const cg1_shuffle_crs* crs_, size_t n_proofs, const uint8_t* instances, const uint8_t* proofs,
                              uint8_t* out_points48) {
  if (!crs_ || (n_proofs && (!instances || !proofs || !out_points48))) return CG1_ERR_ARG;
  const Crs& crs = *reinterpret_cast<const Crs*>(crs_);
  const Layout L(crs.ell, crs.lg);
  const size_t inst_b = 4 * crs.ell * 48, proof_b = proof_wire_bytes(crs.lg), lg = crs.lg;
  for (size_t i = 0; i < n_proofs; ++i) {
    uint8_t* o = out_points48 + i * L.count() * 48;
    memcpy(o, instances + i * inst_b, inst_b);
    o += inst_b;
    const uint8_t* p = proofs + i * proof_b;
    const size_t runs[5] = {10, 2 + 4 * lg, 4, 3 + 6 * lg, 0}, skips[5] = {1, 2, 3, 1, 0};    // points, then Fr fields
    for (int k = 0; k < 4; ++k) {
      memcpy(o, p, 48 * runs[k]);
      o += 48 * runs[k];
      p += 48 * runs[k] + 32 * skips[k];
    }
  }
  return CG1_OK;
}

// What the device front-end needs besides the points (csrc/kernels_frontend.h): per proof the seven Fr fields of the wire proof in
// wire order (r_p c_final d_final z_k z_t z_u x_final) followed by the proof's 12 weights, 19 x 32 bytes.
int cg1_shuffle_gather_aux(const cg1_shuffle_crs* crs_, size_t n_proofs, const uint8_t* proofs, const uint8_t* weights, uint8_t* out_aux) {
  if (!crs_ || (n_proofs && (!proofs || !weights || !out_aux))) return CG1_ERR_ARG;
  const Crs& crs = *reinterpret_cast<const Crs*>(crs_);
  const size_t lg = crs.lg, proof_b = proof_wire_bytes(lg);
  const size_t off_rp = 48 * 10, off_cd = off_rp + 32 + 48 * (2 + 4 * lg), off_z = off_cd + 64 + 48 * 4, off_x = off_z + 96 + 48 * (3 + 6 * lg);
  for (size_t i = 0; i < n_proofs; ++i) {
    const uint8_t* p = proofs + i * proof_b;
    uint8_t* o = out_aux + i * 19 * 32;
    memcpy(o, p + off_rp, 32);
    memcpy(o + 32, p + off_cd, 64);
    memcpy(o + 96, p + off_z, 96);
    memcpy(o + 192, p + off_x, 32);
    memcpy(o + 224, weights + i * 12 * 32, 12 * 32);
  }
  return CG1_OK;
}

// Fold the GPU decompression verdicts (one status byte per own point) into the per-proof status: a proof with any
// undecodable point is rejected (BufReader.read_g1 raises, util.py:143-147) and its scalars are zeroed.
int cg1_shuffle_apply_point_status(int32_t* status, const uint8_t* point_status, size_t n_proofs, size_t points_per_proof,
                                   uint8_t* scalars32, uint8_t* crs_scalars32, size_t ncrs) {
  if (n_proofs && (!status || !point_status || !scalars32 || !crs_scalars32)) return CG1_ERR_ARG;
  for (size_t i = 0; i < n_proofs; ++i) {
    if (status[i]) continue;
    const uint8_t* ps = point_status + i * points_per_proof;
    uint8_t any = 0;
    for (size_t k = 0; k < points_per_proof; ++k) any |= ps[k];
    if (any) {
      status[i] = CG1_SHUFFLE_BAD_POINT;
      memset(scalars32 + i * points_per_proof * 32, 0, points_per_proof * 32);
      memset(crs_scalars32 + i * ncrs * 32, 0, ncrs * 32);
    }
  }
  return CG1_OK;
}

// out[j] = sum over the proofs with status[i] == 0 of crs_scalars[i][j]   (mod r)
int cg1_shuffle_sum_crs_scalars(const uint8_t* crs_scalars32, const int32_t* status, size_t n_proofs, size_t ncrs, uint8_t* out32) {
  std::vector<fr> acc(ncrs, fr_zero());
  for (size_t i = 0; i < n_proofs; ++i) {
    if (status && status[i]) continue;
    for (size_t j = 0; j < ncrs; ++j) {
      fr v;
      memcpy(v.l, crs_scalars32 + (i * ncrs + j) * 32, 32);       // plain (non-Montgomery) canonical values add the same way
      if (geq_r(v.l)) return CG1_ERR_ARG;
      acc[j] = fr_add(acc[j], v);
    }
  }
  for (size_t j = 0; j < ncrs; ++j) memcpy(out32 + 32 * j, acc[j].l, 32);
  return CG1_OK;
}

}  // extern "C"
