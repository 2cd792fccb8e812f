// Host-side BLS12-381 Fp / G1.  See host_g1.h for the role of this file.
#include "host_g1.h"
#include "bls_consts.h"
#include "fe_mul_x86.h"
#include <cstring>
#include <vector>

namespace cg1h {

typedef unsigned __int128 u128;
using cg1::H_P; using cg1::H_PINV; using cg1::H_R1; using cg1::H_R2;

static inline fe mk(const uint64_t* w) { fe r; for (int i = 0; i < 6; ++i) r.l[i] = w[i]; return r; }

fe fe_zero() { fe r; memset(&r, 0, sizeof r); return r; }
fe fe_one() { return mk(H_R1); }
bool fe_is_zero(const fe& a) { uint64_t o = 0; for (int i = 0; i < 6; ++i) o |= a.l[i]; return o == 0; }
bool fe_eq(const fe& a, const fe& b) { uint64_t o = 0; for (int i = 0; i < 6; ++i) o |= a.l[i] ^ b.l[i]; return o == 0; }

static inline bool geq_p(const uint64_t* a) {
  for (int i = 5; i >= 0; --i) { if (a[i] > H_P[i]) return true; if (a[i] < H_P[i]) return false; }
  return true;
}
static inline void sub_p(uint64_t* a) {
  u128 br = 0;
  for (int i = 0; i < 6; ++i) { u128 t = (u128)a[i] - H_P[i] - br; a[i] = (uint64_t)t; br = (t >> 64) & 1; }
}

bool fe_is_canonical(const fe& a) { return !geq_p(a.l); }

fe fe_add(const fe& a, const fe& b) {
  fe r; u128 c = 0;
  for (int i = 0; i < 6; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
  // p < 2^381 so a+b < 2^382 never carries out of 384 bits
  if (geq_p(r.l)) sub_p(r.l);
  return r;
}
fe fe_sub(const fe& a, const fe& b) {
  fe r; u128 br = 0;
  for (int i = 0; i < 6; ++i) { u128 t = (u128)a.l[i] - b.l[i] - br; r.l[i] = (uint64_t)t; br = (t >> 64) & 1; }
  if (br) { u128 c = 0; for (int i = 0; i < 6; ++i) { c += (u128)r.l[i] + H_P[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
  return r;
}
fe fe_neg(const fe& a) { return fe_is_zero(a) ? a : fe_sub(fe_zero(), a); }

// CIOS with the "no-carry" shortcut: p < 2^381 leaves the top word of every partial sum below 2^63, so the
// running value never needs a 7th word beyond one carry word (the method used by gnark/arkworks for moduli with
// a free top bit).  Fully unrolled by the compiler (fixed trip counts).
#if defined(__x86_64__)
// mulx + adcx / adox when the CPU has them (fe_mul_x86.h, generated): the product's two carry chains per row run side by side
static const bool g_has_adx = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("adx");
#endif

fe fe_mul(const fe& a, const fe& b) {
#if defined(__x86_64__)
  if (g_has_adx) {
    fe r;
    fe_mul_adx(r.l, a.l, b.l, H_P, H_PINV);
    if (geq_p(r.l)) sub_p(r.l);
    return r;
  }
#endif
  uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
#pragma GCC unroll 6
  for (int i = 0; i < 6; ++i) {
    const uint64_t bi = b.l[i];
    u128 c = (u128)a.l[0] * bi + t0;
    const uint64_t m = (uint64_t)c * H_PINV;
    u128 d = (u128)m * H_P[0] + (uint64_t)c;          // low word becomes 0
    c >>= 64; d >>= 64;
    c += (u128)a.l[1] * bi + t1; d += (u128)m * H_P[1] + (uint64_t)c; t0 = (uint64_t)d; c >>= 64; d >>= 64;
    c += (u128)a.l[2] * bi + t2; d += (u128)m * H_P[2] + (uint64_t)c; t1 = (uint64_t)d; c >>= 64; d >>= 64;
    c += (u128)a.l[3] * bi + t3; d += (u128)m * H_P[3] + (uint64_t)c; t2 = (uint64_t)d; c >>= 64; d >>= 64;
    c += (u128)a.l[4] * bi + t4; d += (u128)m * H_P[4] + (uint64_t)c; t3 = (uint64_t)d; c >>= 64; d >>= 64;
    c += (u128)a.l[5] * bi + t5; d += (u128)m * H_P[5] + (uint64_t)c; t4 = (uint64_t)d; c >>= 64; d >>= 64;
    t5 = (uint64_t)c + (uint64_t)d;                   // fits: p < 2^381
  }
  fe r; r.l[0] = t0; r.l[1] = t1; r.l[2] = t2; r.l[3] = t3; r.l[4] = t4; r.l[5] = t5;
  if (geq_p(r.l)) sub_p(r.l);
  return r;
}
fe fe_sqr(const fe& a) {
#if defined(__x86_64__)
  if (g_has_adx) {                                     // 57 mulx instead of the product's 72
    fe r;
    fe_sqr_adx(r.l, a.l, H_P, H_PINV);
    if (geq_p(r.l)) sub_p(r.l);
    return r;
  }
#endif
  return fe_mul(a, a);
}

// a^e, fixed 4-bit windows over the (public, constant) exponents of the inversion and the square root: 14 products for the table,
// then 4 squarings + at most one product per nibble -- ~380 squarings + ~100 products instead of the ~570 operations of the bit-by-bit
// ladder (a decompression, util.py:35-36, is one such exponentiation; a compression one inversion)
static fe fe_pow(const fe& a, const uint64_t e[6]) {
  fe tab[16];
  tab[1] = a;
  for (int i = 2; i < 16; ++i) tab[i] = (i & 1) ? fe_mul(tab[i - 1], a) : fe_sqr(tab[i / 2]);
  fe r = fe_one();
  bool started = false;
  for (int nib = 95; nib >= 0; --nib) {
    const unsigned d = (unsigned)(e[nib >> 4] >> ((nib & 15) * 4)) & 15u;
    if (started) { r = fe_sqr(r); r = fe_sqr(r); r = fe_sqr(r); r = fe_sqr(r); }
    if (d) { r = started ? fe_mul(r, tab[d]) : tab[d]; started = true; }
  }
  return started ? r : fe_one();
}
fe fe_inv(const fe& a) { return fe_pow(a, cg1::H_INV_EXP); }
bool fe_sqrt(const fe& a, fe& out) {
  fe s = fe_pow(a, cg1::H_SQRT_EXP);     // p = 3 mod 4
  if (!fe_eq(fe_sqr(s), a)) return false;
  out = s;
  return true;
}

fe fe_from_std(const uint64_t w[6]) { return fe_mul(mk(w), mk(H_R2)); }
void fe_to_std(const fe& a, uint64_t w[6]) {
  fe one = fe_zero(); one.l[0] = 1;
  fe r = fe_mul(a, one);
  for (int i = 0; i < 6; ++i) w[i] = r.l[i];
}
bool fe_from_le48(const uint8_t* b, fe& out) {
  uint64_t w[6];
  for (int i = 0; i < 6; ++i) { uint64_t v = 0; for (int j = 7; j >= 0; --j) v = (v << 8) | b[8 * i + j]; w[i] = v; }
  if (geq_p(w)) return false;
  out = fe_from_std(w);
  return true;
}
void fe_to_le48(const fe& a, uint8_t* b) {
  uint64_t w[6]; fe_to_std(a, w);
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) b[8 * i + j] = (uint8_t)(w[i] >> (8 * j));
}
bool fe_from_be48(const uint8_t* b, fe& out) {
  uint8_t le[48];
  for (int i = 0; i < 48; ++i) le[i] = b[47 - i];
  return fe_from_le48(le, out);
}
void fe_to_be48(const fe& a, uint8_t* b) {
  uint8_t le[48]; fe_to_le48(a, le);
  for (int i = 0; i < 48; ++i) b[i] = le[47 - i];
}
bool fe_lex_largest(const fe& a) {
  uint64_t w[6]; fe_to_std(a, w);
  for (int i = 5; i >= 0; --i) {
    if (w[i] > cg1::H_P_MINUS_1_HALF[i]) return true;
    if (w[i] < cg1::H_P_MINUS_1_HALF[i]) return false;
  }
  return false;
}

// ------------------------------------------------------------------ group law (Jacobian, a = 0)
jac jac_identity() { jac r; r.X = fe_one(); r.Y = fe_one(); r.Z = fe_zero(); return r; }
jac jac_generator() { jac r; r.X = mk(cg1::H_GX); r.Y = mk(cg1::H_GY); r.Z = fe_one(); return r; }
bool jac_is_identity(const jac& a) { return fe_is_zero(a.Z); }

jac jac_dbl(const jac& p) {
  if (jac_is_identity(p)) return p;
  // dbl-2009-l
  fe A = fe_sqr(p.X), B = fe_sqr(p.Y), C = fe_sqr(B);
  fe t = fe_add(p.X, B);
  fe D = fe_sub(fe_sub(fe_sqr(t), A), C); D = fe_add(D, D);
  fe E = fe_add(fe_add(A, A), A);
  fe F = fe_sqr(E);
  jac r;
  r.X = fe_sub(F, fe_add(D, D));
  fe C8 = fe_add(C, C); C8 = fe_add(C8, C8); C8 = fe_add(C8, C8);
  r.Y = fe_sub(fe_mul(E, fe_sub(D, r.X)), C8);
  fe yz = fe_mul(p.Y, p.Z);
  r.Z = fe_add(yz, yz);
  return r;
}

jac jac_add(const jac& p, const jac& q) {
  if (jac_is_identity(p)) return q;
  if (jac_is_identity(q)) return p;
  fe Z1Z1 = fe_sqr(p.Z), Z2Z2 = fe_sqr(q.Z);
  fe U1 = fe_mul(p.X, Z2Z2), U2 = fe_mul(q.X, Z1Z1);
  fe S1 = fe_mul(fe_mul(p.Y, q.Z), Z2Z2), S2 = fe_mul(fe_mul(q.Y, p.Z), Z1Z1);
  if (fe_eq(U1, U2)) {
    if (fe_eq(S1, S2)) return jac_dbl(p);
    return jac_identity();
  }
  fe H = fe_sub(U2, U1), Rr = fe_sub(S2, S1);
  fe HH = fe_sqr(H), HHH = fe_mul(H, HH), V = fe_mul(U1, HH);
  jac r;
  r.X = fe_sub(fe_sub(fe_sqr(Rr), HHH), fe_add(V, V));
  r.Y = fe_sub(fe_mul(Rr, fe_sub(V, r.X)), fe_mul(S1, HHH));
  r.Z = fe_mul(fe_mul(p.Z, q.Z), H);
  return r;
}

// p + (x2, y2) with the second operand affine and not the identity (7M + 4S instead of 11M + 5S)
jac jac_madd(const jac& p, const fe& x2, const fe& y2) {
  if (jac_is_identity(p)) return jac_from_affine(x2, y2);
  fe Z1Z1 = fe_sqr(p.Z);
  fe U2 = fe_mul(x2, Z1Z1), S2 = fe_mul(fe_mul(y2, p.Z), Z1Z1);
  if (fe_eq(p.X, U2)) {
    if (fe_eq(p.Y, S2)) return jac_dbl(p);
    return jac_identity();
  }
  fe H = fe_sub(U2, p.X), Rr = fe_sub(S2, p.Y);
  fe HH = fe_sqr(H), HHH = fe_mul(H, HH), V = fe_mul(p.X, HH);
  jac r;
  r.X = fe_sub(fe_sub(fe_sqr(Rr), HHH), fe_add(V, V));
  r.Y = fe_sub(fe_mul(Rr, fe_sub(V, r.X)), fe_mul(p.Y, HHH));
  r.Z = fe_mul(p.Z, H);
  return r;
}

jac jac_neg(const jac& a) { jac r = a; r.Y = fe_neg(a.Y); return r; }

bool jac_eq(const jac& a, const jac& b) {
  bool ai = jac_is_identity(a), bi = jac_is_identity(b);
  if (ai || bi) return ai && bi;
  fe Z1Z1 = fe_sqr(a.Z), Z2Z2 = fe_sqr(b.Z);
  if (!fe_eq(fe_mul(a.X, Z2Z2), fe_mul(b.X, Z1Z1))) return false;
  return fe_eq(fe_mul(fe_mul(a.Y, b.Z), Z2Z2), fe_mul(fe_mul(b.Y, a.Z), Z1Z1));
}

jac jac_mul(const jac& a, const uint8_t k[32]) {
  // width-5 non-adjacent form over the odd multiples {1, 3, ..., 15} a: one doubling + 7 additions for the table, then one doubling per
  // scalar bit and an addition for every ~6th of them (43 for a 255-bit scalar, against the 60 + 14 of fixed 4-bit windows)
  if (jac_is_identity(a)) return a;
  uint64_t w[5] = {0, 0, 0, 0, 0};                       // the scalar, with room for the recoding carry
  for (int i = 0; i < 32; ++i) w[i >> 3] |= (uint64_t)k[i] << (8 * (i & 7));
  int8_t naf[258];
  int len = 0;
  while (w[0] | w[1] | w[2] | w[3] | w[4]) {
    int d = 0;
    if (w[0] & 1) {
      d = (int)(w[0] & 31);
      if (d > 16) d -= 32;
      if (d > 0) {                                         // w -= d
        uint64_t b = (uint64_t)d;
        for (int i = 0; i < 5 && b; ++i) { const uint64_t t = w[i]; w[i] = t - b; b = t < b ? 1 : 0; }
      } else {                                             // w += -d
        uint64_t c = (uint64_t)(-d);
        for (int i = 0; i < 5 && c; ++i) { const uint64_t t = w[i] + c; c = t < c ? 1 : 0; w[i] = t; }
      }
    }
    naf[len++] = (int8_t)d;
    for (int i = 0; i < 4; ++i) w[i] = (w[i] >> 1) | (w[i + 1] << 63);
    w[4] >>= 1;
  }
  jac tab[8];                                            // tab[j] = (2j + 1) a
  tab[0] = a;
  const jac a2 = jac_dbl(a);
  for (int j = 1; j < 8; ++j) tab[j] = jac_add(tab[j - 1], a2);
  jac acc = jac_identity();
  for (int i = len - 1; i >= 0; --i) {
    acc = jac_dbl(acc);
    const int d = naf[i];
    if (d > 0) acc = jac_add(acc, tab[d >> 1]);
    else if (d < 0) acc = jac_add(acc, jac_neg(tab[(-d) >> 1]));
  }
  return acc;
}

void jac_to_affine(const jac& a, fe& x, fe& y, bool& inf) {
  inf = jac_is_identity(a);
  if (inf) { x = fe_zero(); y = fe_zero(); return; }
  if (fe_eq(a.Z, fe_one())) { x = a.X; y = a.Y; return; }       // a normal form already (decoded points, the generator): no inversion
  fe zi = fe_inv(a.Z), zi2 = fe_sqr(zi);
  x = fe_mul(a.X, zi2);
  y = fe_mul(a.Y, fe_mul(zi2, zi));
}
jac jac_from_affine(const fe& x, const fe& y) { jac r; r.X = x; r.Y = y; r.Z = fe_one(); return r; }
jac jac_from_xyzz(const fe& X, const fe& Y, const fe& ZZ, const fe& ZZZ) {
  // x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2.  Take Z := ZZ: Z^2 = ZZ^2, Z^3 = ZZ^3 = ZZZ^2.
  jac r; r.X = fe_mul(X, ZZ); r.Y = fe_mul(Y, ZZZ); r.Z = ZZ; return r;
}

bool jac_on_curve(const jac& a) {
  if (jac_is_identity(a)) return true;
  // Y^2 = X^3 + 4 Z^6
  fe z2 = fe_sqr(a.Z), z6 = fe_mul(fe_sqr(z2), z2);
  return fe_eq(fe_sqr(a.Y), fe_add(fe_mul(fe_sqr(a.X), a.X), fe_mul(mk(cg1::H_B4), z6)));
}
bool jac_in_subgroup(const jac& a) {
  uint8_t r[32];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) r[8 * i + j] = (uint8_t)(cg1::H_FR[i] >> (8 * j));
  return jac_is_identity(jac_mul(a, r));
}

void jac_batch_to_affine(const jac* pts, size_t n, fe* xs, fe* ys, uint8_t* inf) {
  // Montgomery's trick: one inversion for the whole batch
  std::vector<fe> pref(n);
  fe acc = fe_one();
  for (size_t i = 0; i < n; ++i) {
    inf[i] = jac_is_identity(pts[i]) ? 1 : 0;
    pref[i] = acc;
    if (!inf[i]) acc = fe_mul(acc, pts[i].Z);
  }
  fe inv = fe_inv(acc);
  for (size_t i = n; i-- > 0;) {
    if (inf[i]) { xs[i] = fe_zero(); ys[i] = fe_zero(); continue; }
    fe zi = fe_mul(inv, pref[i]);
    inv = fe_mul(inv, pts[i].Z);
    fe zi2 = fe_sqr(zi);
    xs[i] = fe_mul(pts[i].X, zi2);
    ys[i] = fe_mul(pts[i].Y, fe_mul(zi2, zi));
  }
}

void g1_compress_affine(const fe& x, const fe& y, bool inf, uint8_t out[48]) {
  if (inf) { memset(out, 0, 48); out[0] = 0xC0; return; }
  fe_to_be48(x, out);
  out[0] |= 0x80;
  if (fe_lex_largest(y)) out[0] |= 0x20;
}

void g1_compress(const jac& a, uint8_t out[48]) {
  fe x, y; bool inf;
  jac_to_affine(a, x, y, inf);
  if (inf) { memset(out, 0, 48); out[0] = 0xC0; return; }
  fe_to_be48(x, out);
  out[0] |= 0x80;
  if (fe_lex_largest(y)) out[0] |= 0x20;
}

int g1_decompress(const uint8_t in[48], bool check_subgroup, jac& out) {
  uint8_t flags = in[0];
  bool compressed = flags & 0x80, infinity = flags & 0x40, largest = flags & 0x20;
  if (!compressed) return 1;
  (void)largest;
  // infinity flag: the identity, whatever the sign flag and the remaining bits say -- the wheel (ark-bls12-381 0.4 `read_g1_compressed`) returns the identity as soon as the infinity flag is set, whatever the sign flag and the other 381 bits say, and re-serialises it canonically (0xC0, 47 zero bytes) wherever the reference hashes or compares it; its source is not in /root/reference, so this leniency is restated from the published crate, not pinned by a reference vector
  if (infinity) { out = jac_identity(); return 0; }
  uint8_t xb[48];
  memcpy(xb, in, 48);
  xb[0] &= 0x1F;
  fe x;
  if (!fe_from_be48(xb, x)) return 2;
  fe rhs = fe_add(fe_mul(fe_sqr(x), x), mk(cg1::H_B4));
  fe y;
  if (!fe_sqrt(rhs, y)) return 3;
  if (fe_lex_largest(y) != ((flags & 0x20) != 0)) y = fe_neg(y);
  out = jac_from_affine(x, y);
  if (check_subgroup && !jac_in_subgroup(out)) return 4;
  return 0;
}

}  // namespace cg1h
