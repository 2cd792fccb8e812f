// Host half of the deferred-evaluation G1Point: see lazy_host.h.
#include "lazy_host.h"
#include "bls_consts.h"
#include "pool.h"
#include "../../include/curdle_g1.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <vector>

#ifndef CG1_HD
#define CG1_HD inline
#endif
namespace cg1 {
#include "glv.h"
}

namespace cg1h {

typedef unsigned __int128 u128;

// ------------------------------------------------------------------ Jacobi symbol
// "posdivsteps": the safegcd recurrence with additions instead of subtractions, so f and g stay non-negative and the Jacobi symbol of
// (g / f) can be tracked through it (halving g flips the sign when f = 3, 5 mod 8; exchanging f and g flips it when both are 3 mod 4;
// adding a multiple of f to g changes nothing).  62 steps are run on the low 64 bits of f and g (exact there: after s halvings 64 - s
// low bits of g are still right, and the bookkeeping reads at most 3), collected in a 2 x 2 matrix with entries <= 2^62, which is then
// applied to the 7 x 62-bit limbs.  Every intermediate g is a weighted mean of the previous f and g, so both stay <= p.
// The number of batches has no proven bound for this variant: after MAX_BATCHES the caller decides by Euler's criterion instead
// (never seen in 10^7 random inputs: 381-bit operands finish in 13-17 batches).
namespace {

constexpr int JAC_LIMBS = 7;
constexpr uint64_t M62 = (1ull << 62) - 1;
constexpr int MAX_BATCHES = 40;

struct Trans { uint64_t u, v, q, r; };

inline int64_t posdivsteps_62(int64_t eta, uint64_t f0, uint64_t g0, Trans& t, unsigned& jacp) {
  uint64_t u = 1, v = 0, q = 0, r = 1;
  uint64_t f = f0, g = g0;
  unsigned jac = jacp;
  int i = 62;
  for (;;) {
    const int zeros = __builtin_ctzll(g | (~0ull << i));                // at most i: the sentinel stops the count
    g >>= zeros; u <<= zeros; v <<= zeros;
    eta -= zeros; i -= zeros;
    jac ^= (unsigned)(zeros & ((f >> 1) ^ (f >> 2)));                    // (2 / f) = -1  <=>  f = 3, 5 mod 8
    if (i == 0) break;
    uint64_t w, m;
    int limit;
    if (eta < 0) {
      eta = -eta;
      jac ^= (unsigned)((f & g) >> 1);                                    // reciprocity: both 3 mod 4
      uint64_t tmp;
      tmp = f; f = g; g = tmp;
      tmp = u; u = q; q = tmp;
      tmp = v; v = r; r = tmp;
      limit = ((int)eta + 1) > i ? i : ((int)eta + 1);
      m = (~0ull >> (64 - limit)) & 63u;
      w = (f * g * (f * f - 2)) & m;                                      // -g / f mod 64
    } else {
      limit = ((int)eta + 1) > i ? i : ((int)eta + 1);
      m = (~0ull >> (64 - limit)) & 15u;
      w = f + (((f + 1) & 4) << 1);                                       // 1 / f mod 16
      w = ((0 - w) * g) & m;                                              // -g / f mod 16
    }
    g += f * w; q += u * w; r += v * w;
  }
  t.u = u; t.v = v; t.q = q; t.r = r;
  jacp = jac;
  return eta;
}

inline void update_fg(uint64_t* f, uint64_t* g, const Trans& t, int len) {
  u128 cf = (u128)t.u * f[0] + (u128)t.v * g[0];
  u128 cg = (u128)t.q * f[0] + (u128)t.r * g[0];
  cf >>= 62; cg >>= 62;                                                    // the low 62 bits are zero by construction
  for (int i = 1; i < len; ++i) {
    cf += (u128)t.u * f[i] + (u128)t.v * g[i];
    cg += (u128)t.q * f[i] + (u128)t.r * g[i];
    f[i - 1] = (uint64_t)cf & M62; cf >>= 62;
    g[i - 1] = (uint64_t)cg & M62; cg >>= 62;
  }
  f[len - 1] = (uint64_t)cf;
  g[len - 1] = (uint64_t)cg;
}

// 6 x 64 -> 7 x 62
inline void to62(const uint64_t w[6], uint64_t o[JAC_LIMBS]) {
  for (int i = 0; i < JAC_LIMBS; ++i) {
    const unsigned bit = 62u * (unsigned)i, k = bit >> 6, sh = bit & 63u;
    uint64_t v = k < 6 ? w[k] >> sh : 0;
    if (sh > 2 && k + 1 < 6) v |= w[k + 1] << (64 - sh);
    o[i] = v & M62;
  }
}

// 2: did not finish
int jacobi_words(const uint64_t a[6]) {
  uint64_t f[JAC_LIMBS], g[JAC_LIMBS];
  for (int i = 0; i < JAC_LIMBS; ++i) f[i] = cg1::H_P62[i];
  to62(a, g);
  uint64_t nz = 0;
  for (int i = 0; i < JAC_LIMBS; ++i) nz |= g[i];
  if (!nz) return 0;
  int64_t eta = -1;
  unsigned jac = 0;
  int len = JAC_LIMBS;
  for (int count = 0; count < MAX_BATCHES; ++count) {
    Trans t;
    eta = posdivsteps_62(eta, f[0] | (f[1] << 62), g[0] | (g[1] << 62), t, jac);
    update_fg(f, g, t, len);
    if (f[0] == 1) {                                                       // f == 1: (g / 1) = 1 whatever g still is
      uint64_t rest = 0;
      for (int j = 1; j < len; ++j) rest |= f[j];
      if (!rest) return 1 - 2 * (int)(jac & 1u);
    }
    uint64_t gz = 0;
    for (int j = 0; j < len; ++j) gz |= g[j];
    if (!gz) return 0;                                                     // gcd = f != 1 (cannot happen for 0 < a < p, p prime)
    while (len > 2 && f[len - 1] == 0 && g[len - 1] == 0) --len;
  }
  return 2;
}

}  // namespace

int fe_jacobi(const fe& a) {
  const int j = jacobi_words(a.l);
  if (j != 2) return j;
  // Euler's criterion: a^((p-1)/2)
  if (fe_is_zero(a)) return 0;
  fe s;
  return fe_sqrt(a, s) ? 1 : -1;
}

static inline fe mkc(const uint64_t* w) { fe r; for (int i = 0; i < 6; ++i) r.l[i] = w[i]; return r; }

int g1_validate_compressed(const uint8_t in[48], bool* is_identity) {
  const uint8_t flags = in[0];
  if (is_identity) *is_identity = false;
  if (!(flags & 0x80)) return 1;
  if (flags & 0x40) { if (is_identity) *is_identity = true; return 0; }      // g1_decompress (host_g1.cpp): the identity whatever else is set
  uint8_t xb[48];
  memcpy(xb, in, 48);
  xb[0] &= 0x1F;
  fe x;
  if (!fe_from_be48(xb, x)) return 2;
  const fe rhs = fe_add(fe_mul(fe_sqr(x), x), mkc(cg1::H_B4));
  return fe_jacobi(rhs) < 0 ? 3 : 0;
}

// multiplication by |z| = 0xd201000000010000 (the BLS parameter's absolute value): 63 doublings + 5 additions
static jac mul_zabs(const jac& p) {
  constexpr uint64_t ZABS = 0xd201000000010000ull;
  jac q = p;
  for (int bit = 62; bit >= 0; --bit) {
    q = jac_dbl(q);
    if ((ZABS >> bit) & 1ull) q = jac_add(q, p);
  }
  return q;
}

bool g1_in_subgroup_fast(const fe& x, const fe& y) {
  const jac p = jac_from_affine(x, y);
  jac acc = mul_zabs(mul_zabs(p));                                          // [z^2] P
  const fe yneg = fe_neg(y);
  acc = jac_madd(acc, x, yneg);                                             // - P
  if (jac_is_identity(acc)) return false;                                   // [z^2] P == P and phi(P) == O is impossible for a finite P
  acc = jac_madd(acc, fe_mul(x, mkc(cg1::H_BETA)), yneg);                   // - phi(P)
  return jac_is_identity(acc);
}

// ------------------------------------------------------------------ one linear combination: interleaved width-5 NAF
namespace {

// width-5 NAF digits (odd, |d| <= 15) of a 256-bit little-endian integer; returns the number of digits
int wnaf5(const uint8_t k[32], int8_t naf[260]) {
  uint64_t w[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 32; ++i) w[i >> 3] |= (uint64_t)k[i] << (8 * (i & 7));
  int len = 0;
  while (w[0] | w[1] | w[2] | w[3] | w[4]) {
    int d = 0;
    if (w[0] & 1) {
      d = (int)(w[0] & 31);
      if (d > 16) d -= 32;
      if (d > 0) {
        uint64_t b = (uint64_t)d;
        for (int i = 0; i < 5 && b; ++i) { const uint64_t t = w[i]; w[i] = t - b; b = t < b ? 1 : 0; }
      } else {
        uint64_t c = (uint64_t)(-d);
        for (int i = 0; i < 5 && c; ++i) { const uint64_t t = w[i] + c; c = t < c ? 1 : 0; w[i] = t; }
      }
    }
    naf[len++] = (int8_t)d;
    for (int i = 0; i < 4; ++i) w[i] = (w[i] >> 1) | (w[i + 1] << 63);
    w[4] >>= 1;
  }
  return len;
}

inline bool scalar_is(const uint8_t* s, uint8_t v) {
  if (s[0] != v) return false;
  for (int i = 1; i < 32; ++i) if (s[i]) return false;
  return true;
}

}  // namespace

jac lincomb_one(const aff* pts, const uint32_t* idx, const uint8_t* neg, const uint8_t* scalars32, size_t k) {
  jac units = jac_identity();                       // terms with coefficient 1: plain (mixed) additions
  struct Term { std::vector<int8_t> naf; jac tab[8]; };
  std::vector<Term> terms;
  terms.reserve(k);
  int maxlen = 0;
  for (size_t t = 0; t < k; ++t) {
    const aff& a = pts[idx[t]];
    if (a.inf) continue;
    const uint8_t* s = scalars32 + 32 * t;
    if (scalar_is(s, 0)) continue;
    const fe y = (neg && neg[t]) ? fe_neg(a.y) : a.y;
    if (scalar_is(s, 1)) { units = jac_madd(units, a.x, y); continue; }
    terms.emplace_back();
    Term& T = terms.back();
    int8_t naf[260];
    const int len = wnaf5(s, naf);
    T.naf.assign(naf, naf + len);
    maxlen = std::max(maxlen, len);
    T.tab[0] = jac_from_affine(a.x, y);
    const jac d2 = jac_dbl(T.tab[0]);
    for (int j = 1; j < 8; ++j) T.tab[j] = jac_add(T.tab[j - 1], d2);
  }
  jac acc = jac_identity();
  for (int i = maxlen - 1; i >= 0; --i) {
    acc = jac_dbl(acc);
    for (Term& T : terms) {
      if (i >= (int)T.naf.size()) continue;
      const int d = T.naf[i];
      if (d > 0) acc = jac_add(acc, T.tab[d >> 1]);
      else if (d < 0) acc = jac_add(acc, jac_neg(T.tab[(-d) >> 1]));
    }
  }
  return jac_add(acc, units);
}

static inline bool load_affine96(const uint8_t* rec, aff& o) {
  bool zero = true;
  for (int k = 0; k < 96 && zero; ++k) zero = rec[k] == 0;
  if (zero) { o.inf = true; o.x = fe_zero(); o.y = fe_zero(); return true; }
  o.inf = false;
  return fe_from_le48(rec, o.x) && fe_from_le48(rec + 48, o.y);
}

static inline void run_pool(const std::function<void()>& work, size_t items, int n_threads) {
  cg1::Pool& pool = cg1::Pool::get();
  size_t nt = n_threads > 0 ? (size_t)n_threads : pool.size() + 1;
  nt = std::min(nt, items);
  if (nt <= 1) work(); else pool.run(work, nt);
}

int lincomb_pool_jac(const uint8_t* bases_affine96, size_t n_bases, const uint32_t* offsets, const uint32_t* term_base, const uint8_t* term_scalars32,
                     const uint32_t* sel, size_t n_sel, jac* results, int n_threads) {
  if (n_sel == 0) return 0;
  // only the bases the selected combinations use are converted (a base costs one Montgomery product per coordinate)
  std::vector<aff> pts(n_bases);
  std::vector<uint8_t> loaded(n_bases, 0);
  size_t max_k = 0;
  for (size_t q = 0; q < n_sel; ++q) {
    const size_t j = sel ? sel[q] : q;
    max_k = std::max<size_t>(max_k, offsets[j + 1] - offsets[j]);
    for (size_t t = offsets[j]; t < offsets[j + 1]; ++t) {
      const uint32_t b = term_base[t] & 0x7fffffffu;
      if (b >= n_bases) return 1;
      if (!loaded[b]) { if (!load_affine96(bases_affine96 + 96 * (size_t)b, pts[b])) return 3; loaded[b] = 1; }
    }
  }
  // threads by the work at hand: a thread is worth waking for ~0.1 ms of arithmetic (~400 group operations), not for three additions
  if (n_threads <= 0) {
    double ops = 0;
    for (size_t q = 0; q < n_sel; ++q) {
      const size_t j = sel ? sel[q] : q;
      size_t heavy = 0, unit = 0;
      for (size_t t = offsets[j]; t < offsets[j + 1]; ++t) {
        const uint8_t* sc = term_scalars32 + 32 * t;
        bool small = sc[0] <= 1;
        for (int b = 1; b < 32 && small; ++b) small = sc[b] == 0;
        if (small) ++unit; else ++heavy;
      }
      ops += (heavy ? 255.0 : 0.0) + 52.0 * (double)heavy + (double)unit;
    }
    n_threads = (int)std::max<double>(1.0, std::min<double>(256.0, ops / 400.0));
  }
  std::atomic<size_t> next{0};
  std::function<void()> work = [&]() {
    std::vector<uint32_t> idx(max_k);
    std::vector<uint8_t> neg(max_k);
    for (;;) {
      const size_t q = next.fetch_add(1);
      if (q >= n_sel) return;
      const size_t j = sel ? sel[q] : q;
      const size_t lo = offsets[j], k = offsets[j + 1] - lo;
      for (size_t t = 0; t < k; ++t) { idx[t] = term_base[lo + t] & 0x7fffffffu; neg[t] = (uint8_t)(term_base[lo + t] >> 31); }
      results[j] = lincomb_one(pts.data(), idx.data(), neg.data(), term_scalars32 + 32 * lo, k);
    }
  };
  run_pool(work, n_sel, n_threads);
  return 0;
}

}  // namespace cg1h

// ------------------------------------------------------------------ C ABI (host-only entry points of the deferred evaluation)
namespace {

using cg1h::aff; using cg1h::fe; using cg1h::jac;
using cg1h::load_affine96; using cg1h::run_pool;

}  // namespace

extern "C" void cg1_lincomb_write_outputs(const void* jac_results, size_t n_out, uint8_t* out_blobs144, uint8_t* out_affine96, uint8_t* out_comp48);

extern "C" {

int cg1_validate_compressed(const uint8_t* in48, int* is_identity) {
  if (!in48) return CG1_ERR_ARG;
  bool inf = false;
  const int rc = cg1h::g1_validate_compressed(in48, &inf);
  if (is_identity) *is_identity = inf ? 1 : 0;
  return rc == 0 ? CG1_OK : (rc == 3 ? CG1_ERR_NOT_ON_CURVE : CG1_ERR_ENCODING);
}

int cg1_fp_jacobi(const uint8_t* le48) {
  fe a;
  if (!le48 || !cg1h::fe_from_le48(le48, a)) return 2;
  return cg1h::fe_jacobi(a);
}

int cg1_batch_decompress_pool(const uint8_t* in48, size_t n, uint8_t* out_blobs144, uint8_t* out_affine96, int n_threads, size_t* bad_index) {
  if (n == 0) return CG1_OK;
  if (!in48) return CG1_ERR_ARG;
  std::atomic<size_t> next{0};
  std::atomic<size_t> bad{(size_t)-1};
  std::atomic<int> bad_rc{0};
  const size_t slice = 8, items = (n + slice - 1) / slice;
  if (n_threads <= 0) n_threads = (int)std::max<size_t>(1, std::min<size_t>(256, n / 8));        // ~0.1 ms of square roots per thread
  std::function<void()> work = [&]() {
    for (;;) {
      const size_t it = next.fetch_add(1);
      if (it >= items) return;
      for (size_t i = it * slice; i < std::min(n, it * slice + slice); ++i) {
        jac p;
        const int rc = cg1h::g1_decompress(in48 + 48 * i, false, p);
        if (rc != 0) {
          size_t cur = bad.load();
          while (i < cur && !bad.compare_exchange_weak(cur, i)) {}
          if (bad.load() == i) bad_rc.store(rc == 3 ? CG1_ERR_NOT_ON_CURVE : CG1_ERR_ENCODING);
          p = cg1h::jac_identity();
        }
        if (out_blobs144) memcpy(out_blobs144 + CG1_POINT_BYTES * i, &p, sizeof p);
        if (out_affine96) {
          uint8_t* o = out_affine96 + 96 * i;
          if (cg1h::jac_is_identity(p)) memset(o, 0, 96);
          else { cg1h::fe_to_le48(p.X, o); cg1h::fe_to_le48(p.Y, o + 48); }      // decoded points are in normal form (Z = 1)
        }
      }
    }
  };
  run_pool(work, items, n_threads);
  if (bad.load() != (size_t)-1) { if (bad_index) *bad_index = bad.load(); return bad_rc.load() ? bad_rc.load() : CG1_ERR_ENCODING; }
  return CG1_OK;
}

int cg1_batch_subgroup_pool(const uint8_t* affine96, size_t n, uint8_t* out_flags, int n_threads) {
  if (n == 0) return CG1_OK;
  if (!affine96 || !out_flags) return CG1_ERR_ARG;
  std::atomic<size_t> next{0};
  std::atomic<int> bad{0};
  const size_t slice = 4, items = (n + slice - 1) / slice;
  if (n_threads <= 0) n_threads = (int)std::max<size_t>(1, std::min<size_t>(256, n / 3));        // ~0.1 ms of subgroup tests per thread
  std::function<void()> work = [&]() {
    for (;;) {
      const size_t it = next.fetch_add(1);
      if (it >= items) return;
      for (size_t i = it * slice; i < std::min(n, it * slice + slice); ++i) {
        aff a;
        if (!load_affine96(affine96 + 96 * i, a)) { bad.store(1); out_flags[i] = 0; continue; }
        out_flags[i] = a.inf ? 1 : (cg1h::g1_in_subgroup_fast(a.x, a.y) ? 1 : 0);
      }
    }
  };
  run_pool(work, items, n_threads);
  return bad.load() ? CG1_ERR_ENCODING : CG1_OK;
}

// The pool path of cg1_lincomb_batch (msm_gpu.hip chooses between this and the GPU's batched MSM): see include/curdle_g1.h.
int cg1_lincomb_batch_pool(const uint8_t* bases_affine96, size_t n_bases, const uint32_t* offsets, size_t n_out, const uint32_t* term_base,
                           const uint8_t* term_scalars32, uint8_t* out_blobs144, uint8_t* out_affine96, uint8_t* out_comp48, int n_threads) {
  if (n_out == 0) return CG1_OK;
  if (!offsets || offsets[0] != 0) return CG1_ERR_ARG;
  const size_t T = offsets[n_out];
  if (T && (!bases_affine96 || !term_base || !term_scalars32)) return CG1_ERR_ARG;
  for (size_t j = 0; j < n_out; ++j) if (offsets[j] > offsets[j + 1]) return CG1_ERR_ARG;
  std::vector<jac> res(n_out);
  const int rc = cg1h::lincomb_pool_jac(bases_affine96, n_bases, offsets, term_base, term_scalars32, nullptr, n_out, res.data(), n_threads);
  if (rc) return rc == 3 ? CG1_ERR_ENCODING : CG1_ERR_ARG;
  cg1_lincomb_write_outputs(res.data(), n_out, out_blobs144, out_affine96, out_comp48);
  return CG1_OK;
}

// n results -> normalised blobs (Z = 1) / affine96 / compressed48 (each may be NULL) with ONE shared inversion
void cg1_lincomb_write_outputs(const void* jac_results, size_t n_out, uint8_t* out_blobs144, uint8_t* out_affine96, uint8_t* out_comp48) {
  const jac* res = static_cast<const jac*>(jac_results);
  std::vector<fe> xs(n_out), ys(n_out);
  std::vector<uint8_t> inf(n_out);
  cg1h::jac_batch_to_affine(res, n_out, xs.data(), ys.data(), inf.data());
  for (size_t j = 0; j < n_out; ++j) {
    if (out_blobs144) {
      const jac p = inf[j] ? cg1h::jac_identity() : cg1h::jac_from_affine(xs[j], ys[j]);
      memcpy(out_blobs144 + CG1_POINT_BYTES * j, &p, sizeof p);
    }
    if (out_affine96) {
      uint8_t* o = out_affine96 + 96 * j;
      if (inf[j]) memset(o, 0, 96);
      else { cg1h::fe_to_le48(xs[j], o); cg1h::fe_to_le48(ys[j], o + 48); }
    }
    if (out_comp48) cg1h::g1_compress_affine(xs[j], ys[j], inf[j] != 0, out_comp48 + 48 * j);
  }
}

// The endomorphism split the device digit kernels use (csrc/glv.h), one scalar: for the CPU tests.
void cg1_glv_split(const uint8_t* scalar32, uint8_t* k1_16, uint8_t* k2_16, int* neg1, int* neg2) {
  uint32_t k[8];
  memcpy(k, scalar32, 32);
  cg1::GlvParts o;
  cg1::glv_split(k, o);
  memcpy(k1_16, o.k1, 16);
  memcpy(k2_16, o.k2, 16);
  *neg1 = (int)o.neg1; *neg2 = (int)o.neg2;
}

}  // extern "C"
