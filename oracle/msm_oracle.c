/* CPU ORACLE (test infrastructure, NOT product code) -- plain C restatement of the reference's MSM.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Restates /root/reference/curdleproofs/curdleproofs/msm_accumulator.py:6-12:
 *
 *     current = G1Point.identity()                    (:9)
 *     for (base, scalar) in zip(bases, scalars):      (:10)
 *         current = current + (base * scalar)         (:11)   one full-width scalar-mul, then one add
 *     return current                                  (:12)
 *
 * `base * scalar` lives in the third-party wheel py_arkworks_bls12381 0.3.5 (curdleproofs/pyproject.toml:10;
 * source not under /root/reference).  Its published algorithm is a plain MSB-first double-and-add over
 * the 255-bit scalar in Jacobian coordinates; that is what orc_scalar_mul does.  Single thread, like the
 * reference.  Parity pinning: checked against oracle/bls12_381.py and the reference's KATs
 * (generator, 99*G: test_curdleproofs.py:179-180, :233-236) in tests/test_oracle_kat.py.
 *
 * Deliberately written differently from the product's host code (separated schoolbook product + REDC
 * instead of CIOS; bit-serial ladder instead of windows) so the two do not share a bug.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[6]; } fq;        /* Montgomery form, radix 2^384, canonical */
typedef struct { fq x, y, z; } pt;            /* Jacobian; z == 0 is the identity */

static const uint64_t PRIME[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                  0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t NPRIME = 0x89f3fffcfffcfffdULL;                    /* -p^-1 mod 2^64 */
static const uint64_t RR[6] = {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                               0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};   /* 2^768 mod p */

static int ge_p(const uint64_t* a) {
  for (int i = 5; i >= 0; --i) { if (a[i] != PRIME[i]) return a[i] > PRIME[i]; }
  return 1;
}
static void minus_p(uint64_t* a) {
  uint64_t borrow = 0;
  for (int i = 0; i < 6; ++i) {
    u128 d = (u128)a[i] - PRIME[i] - borrow;
    a[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
  }
}

static fq fq_add(fq a, fq b) {
  fq r; uint64_t c = 0;
  for (int i = 0; i < 6; ++i) { u128 s = (u128)a.v[i] + b.v[i] + c; r.v[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
  if (ge_p(r.v)) minus_p(r.v);
  return r;
}
static fq fq_sub(fq a, fq b) {
  fq r; uint64_t borrow = 0;
  for (int i = 0; i < 6; ++i) { u128 d = (u128)a.v[i] - b.v[i] - borrow; r.v[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1; }
  if (borrow) { uint64_t c = 0; for (int i = 0; i < 6; ++i) { u128 s = (u128)r.v[i] + PRIME[i] + c; r.v[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } }
  return r;
}
/* schoolbook 6x6 -> 12 words, then word-by-word Montgomery reduction (SOS) */
static fq fq_mul(fq a, fq b) {
  uint64_t t[13];
  memset(t, 0, sizeof t);
  for (int i = 0; i < 6; ++i) {
    uint64_t c = 0;
    for (int j = 0; j < 6; ++j) { u128 s = (u128)a.v[i] * b.v[j] + t[i + j] + c; t[i + j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    t[i + 6] = c;
  }
  for (int i = 0; i < 6; ++i) {
    uint64_t m = t[i] * NPRIME, c = 0;
    for (int j = 0; j < 6; ++j) { u128 s = (u128)m * PRIME[j] + t[i + j] + c; t[i + j] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    for (int k = i + 6; c && k < 13; ++k) { u128 s = (u128)t[k] + c; t[k] = (uint64_t)s; c = (uint64_t)(s >> 64); }
  }
  fq r;
  for (int i = 0; i < 6; ++i) r.v[i] = t[6 + i];
  if (t[12] || ge_p(r.v)) minus_p(r.v);
  return r;
}
static int fq_is_zero(fq a) { uint64_t o = 0; for (int i = 0; i < 6; ++i) o |= a.v[i]; return o == 0; }
static int fq_equal(fq a, fq b) { uint64_t o = 0; for (int i = 0; i < 6; ++i) o |= a.v[i] ^ b.v[i]; return o == 0; }
static fq fq_from_le(const uint8_t* b) {
  fq r;
  for (int i = 0; i < 6; ++i) { uint64_t w = 0; for (int j = 7; j >= 0; --j) w = (w << 8) | b[8 * i + j]; r.v[i] = w; }
  fq rr; memcpy(rr.v, RR, sizeof RR);
  return fq_mul(r, rr);
}
static void fq_to_le(fq a, uint8_t* b) {
  fq one; memset(&one, 0, sizeof one); one.v[0] = 1;
  fq s = fq_mul(a, one);
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 8; ++j) b[8 * i + j] = (uint8_t)(s.v[i] >> (8 * j));
}
static fq fq_pow_pm2(fq a) {   /* a^(p-2) */
  uint64_t e[6]; memcpy(e, PRIME, sizeof e); e[0] -= 2;
  fq one; memset(&one, 0, sizeof one); one.v[0] = 1;
  fq rr; memcpy(rr.v, RR, sizeof RR);
  fq r = fq_mul(one, rr);
  for (int i = 383; i >= 0; --i) { r = fq_mul(r, r); if ((e[i >> 6] >> (i & 63)) & 1) r = fq_mul(r, a); }
  return r;
}

static pt pt_identity(void) { pt r; memset(&r, 0, sizeof r); return r; }
static pt pt_double(pt p) {
  if (fq_is_zero(p.z)) return p;
  /* x3 = (3x^2)^2 - 8xy^2 ; y3 = 3x^2 (4xy^2 - x3) - 8y^4 ; z3 = 2yz   (a = 0) */
  fq xx = fq_mul(p.x, p.x), yy = fq_mul(p.y, p.y), yyyy = fq_mul(yy, yy);
  fq s = fq_mul(p.x, yy); s = fq_add(s, s); s = fq_add(s, s);            /* 4xy^2 */
  fq m = fq_add(fq_add(xx, xx), xx);
  pt r;
  r.x = fq_sub(fq_mul(m, m), fq_add(s, s));
  fq y8 = fq_add(yyyy, yyyy); y8 = fq_add(y8, y8); y8 = fq_add(y8, y8);
  r.y = fq_sub(fq_mul(m, fq_sub(s, r.x)), y8);
  fq yz = fq_mul(p.y, p.z);
  r.z = fq_add(yz, yz);
  return r;
}
static pt pt_add(pt p, pt q) {
  if (fq_is_zero(p.z)) return q;
  if (fq_is_zero(q.z)) return p;
  fq z1z1 = fq_mul(p.z, p.z), z2z2 = fq_mul(q.z, q.z);
  fq u1 = fq_mul(p.x, z2z2), u2 = fq_mul(q.x, z1z1);
  fq s1 = fq_mul(p.y, fq_mul(q.z, z2z2)), s2 = fq_mul(q.y, fq_mul(p.z, z1z1));
  if (fq_equal(u1, u2)) return fq_equal(s1, s2) ? pt_double(p) : pt_identity();
  fq h = fq_sub(u2, u1), r = fq_sub(s2, s1);
  fq hh = fq_mul(h, h), hhh = fq_mul(h, hh), v = fq_mul(u1, hh);
  pt o;
  o.x = fq_sub(fq_sub(fq_mul(r, r), hhh), fq_add(v, v));
  o.y = fq_sub(fq_mul(r, fq_sub(v, o.x)), fq_mul(s1, hhh));
  o.z = fq_mul(h, fq_mul(p.z, q.z));
  return o;
}
static pt pt_from_affine96(const uint8_t* b) {
  int any = 0;
  for (int i = 0; i < 96; ++i) any |= b[i];
  if (!any) return pt_identity();
  pt r;
  r.x = fq_from_le(b); r.y = fq_from_le(b + 48);
  uint8_t one[48]; memset(one, 0, sizeof one); one[0] = 1;
  r.z = fq_from_le(one);
  return r;
}
static void pt_to_affine96(pt p, uint8_t* out) {
  if (fq_is_zero(p.z)) { memset(out, 0, 96); return; }
  fq zi = fq_pow_pm2(p.z), zi2 = fq_mul(zi, zi);
  fq_to_le(fq_mul(p.x, zi2), out);
  fq_to_le(fq_mul(p.y, fq_mul(zi2, zi)), out + 48);
}
/* base * scalar: MSB-first double-and-add over all 256 bits of the little-endian scalar */
static pt pt_scalar_mul(pt base, const uint8_t* k) {
  pt acc = pt_identity();
  for (int bit = 255; bit >= 0; --bit) {
    acc = pt_double(acc);
    if ((k[bit >> 3] >> (bit & 7)) & 1) acc = pt_add(acc, base);
  }
  return acc;
}

/* ---- exported ------------------------------------------------------------------------------ */
/* msm_accumulator.py:6-12.  points: n affine96 records (all-zero = identity); scalars: n x 32 B LE. */
void orc_compute_msm(const uint8_t* points96, const uint8_t* scalars32, size_t n, uint8_t out96[96]) {
  pt current = pt_identity();                                           /* :9  */
  for (size_t i = 0; i < n; ++i)                                        /* :10 */
    current = pt_add(current, pt_scalar_mul(pt_from_affine96(points96 + 96 * i), scalars32 + 32 * i));   /* :11 */
  pt_to_affine96(current, out96);                                       /* :12 */
}
/* Same group element as orc_compute_msm by the textbook bucket method (unsigned c-bit windows, running-sum
 * bucket reduction, Horner over windows).  NOT the reference's algorithm: it exists so that parity tests can
 * compare the GPU at 2^16..2^18 terms in seconds, and as a clearly-labelled stronger CPU baseline in bench.py.
 * Cross-checked against the naive loop in tests/test_oracle_kat.py. */
#include <stdlib.h>
void orc_msm_bucket(const uint8_t* points96, const uint8_t* scalars32, size_t n, int c, uint8_t out96[96]) {
  if (c < 2) c = 2;
  if (c > 20) c = 20;
  const int nwin = (256 + c - 1) / c;
  const size_t nb = (size_t)1 << c;
  pt* pts = (pt*)malloc((n ? n : 1) * sizeof(pt));
  pt* buckets = (pt*)malloc(nb * sizeof(pt));
  for (size_t i = 0; i < n; ++i) pts[i] = pt_from_affine96(points96 + 96 * i);
  pt total = pt_identity();
  for (int w = nwin - 1; w >= 0; --w) {
    for (int k = 0; k < c; ++k) total = pt_double(total);
    for (size_t b = 0; b < nb; ++b) buckets[b] = pt_identity();
    for (size_t i = 0; i < n; ++i) {
      const uint8_t* k = scalars32 + 32 * i;
      uint32_t d = 0;
      for (int t = 0; t < c; ++t) {
        int bit = w * c + t;
        if (bit < 256 && ((k[bit >> 3] >> (bit & 7)) & 1)) d |= 1u << t;
      }
      if (d) buckets[d] = pt_add(buckets[d], pts[i]);
    }
    pt run = pt_identity(), acc = pt_identity();
    for (size_t b = nb - 1; b >= 1; --b) { run = pt_add(run, buckets[b]); acc = pt_add(acc, run); }
    total = pt_add(total, acc);
  }
  pt_to_affine96(total, out96);
  free(pts); free(buckets);
}
/* The same bucket method on `threads` host cores (OpenMP): the (window, point-slice) tasks are independent; their
 * partial window sums are combined by one Horner pass.  A clearly-labelled NON-reference baseline for bench.py
 * ("all host cores"); the reference itself is single-threaded. */
void orc_msm_bucket_mt(const uint8_t* points96, const uint8_t* scalars32, size_t n, int c, int threads, uint8_t out96[96]) {
  if (c < 2) c = 2;
  if (c > 16) c = 16;
  if (threads < 1) threads = 1;
  const int nwin = (256 + c - 1) / c;
  const size_t nb = (size_t)1 << c;
  int nslice = (threads + nwin - 1) / nwin;
  if ((size_t)nslice > n / 1024 + 1) nslice = (int)(n / 1024 + 1);
  const int ntask = nwin * nslice;
  pt* pts = (pt*)malloc((n ? n : 1) * sizeof(pt));
  pt* part = (pt*)malloc((size_t)ntask * sizeof(pt));
#pragma omp parallel for num_threads(threads) schedule(static)
  for (long i = 0; i < (long)n; ++i) pts[i] = pt_from_affine96(points96 + 96 * (size_t)i);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
  for (int t = 0; t < ntask; ++t) {
    const int w = t / nslice, sl = t % nslice;
    const size_t i0 = n * (size_t)sl / (size_t)nslice, i1 = n * (size_t)(sl + 1) / (size_t)nslice;
    pt* buckets = (pt*)malloc(nb * sizeof(pt));
    for (size_t b = 0; b < nb; ++b) buckets[b] = pt_identity();
    for (size_t i = i0; i < i1; ++i) {
      const uint8_t* k = scalars32 + 32 * i;
      uint32_t d = 0;
      for (int u = 0; u < c; ++u) {
        int bit = w * c + u;
        if (bit < 256 && ((k[bit >> 3] >> (bit & 7)) & 1)) d |= 1u << u;
      }
      if (d) buckets[d] = pt_add(buckets[d], pts[i]);
    }
    pt run = pt_identity(), acc = pt_identity();
    for (size_t b = nb - 1; b >= 1; --b) { run = pt_add(run, buckets[b]); acc = pt_add(acc, run); }
    part[t] = acc;
    free(buckets);
  }
  pt total = pt_identity();
  for (int w = nwin - 1; w >= 0; --w) {
    for (int k = 0; k < c; ++k) total = pt_double(total);
    for (int sl = 0; sl < nslice; ++sl) total = pt_add(total, part[w * nslice + sl]);
  }
  pt_to_affine96(total, out96);
  free(pts); free(part);
}
void orc_scalar_mul(const uint8_t* point96, const uint8_t* scalar32, uint8_t out96[96]) {
  pt_to_affine96(pt_scalar_mul(pt_from_affine96(point96), scalar32), out96);
}
void orc_add(const uint8_t* a96, const uint8_t* b96, uint8_t out96[96]) {
  pt_to_affine96(pt_add(pt_from_affine96(a96), pt_from_affine96(b96)), out96);
}
/* 48-byte ZCash-format compression of an affine96 record (G1Point.to_compressed_bytes, util.py:27-28) */
void orc_compress(const uint8_t* a96, uint8_t out48[48]) {
  int any = 0;
  for (int i = 0; i < 96; ++i) any |= a96[i];
  if (!any) { memset(out48, 0, 48); out48[0] = 0xC0; return; }
  for (int i = 0; i < 48; ++i) out48[i] = a96[47 - i];
  /* y > (p-1)/2  <=>  2y > p-1  <=>  2y >= p+1 > p  (p odd) : compare 2y with p as 385-bit integers */
  uint64_t y[6];
  for (int i = 0; i < 6; ++i) { uint64_t w = 0; for (int j = 7; j >= 0; --j) w = (w << 8) | a96[48 + 8 * i + j]; y[i] = w; }
  uint64_t d[7]; uint64_t c = 0;
  for (int i = 0; i < 6; ++i) { d[i] = (y[i] << 1) | c; c = y[i] >> 63; }
  d[6] = c;
  int larger = d[6] != 0;
  if (!larger) {
    larger = 0;
    for (int i = 5; i >= 0; --i) { if (d[i] != PRIME[i]) { larger = d[i] > PRIME[i]; break; } }
  }
  out48[0] |= 0x80;
  if (larger) out48[0] |= 0x20;
}

/* 48-byte ZCash-format decompression: G1Point.from_compressed_bytes_unchecked (util.py:35-36) / from_compressed_bytes
 * (checked, test_curdleproofs.py:171).  Restates oracle/bls12_381.py g1_decompress.  Returns 0 and the affine96
 * record (zeros = identity), or 1 bad encoding / 2 x not on the curve / 3 not in the prime-order subgroup. */
static const uint64_t SQRT_EXP[6] = {0xee7fbfffffffeaabULL, 0x07aaffffac54ffffULL, 0xd9cc34a83dac3d89ULL,
                                     0xd91dd2e13ce144afULL, 0x92c6e9ed90d2eb35ULL, 0x0680447a8e5ff9a6ULL};   /* (p+1)/4 */
static const uint64_t GROUP_ORDER[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
int orc_decompress(const uint8_t in48[48], int check_subgroup, uint8_t out96[96]) {
  memset(out96, 0, 96);
  const uint8_t flags = in48[0];
  if (!(flags & 0x80)) return 1;
  if (flags & 0x40) return 0;              /* the identity whatever the other bits say (oracle/bls12_381.py g1_decompress) */
  uint8_t xle[48];
  for (int i = 0; i < 48; ++i) xle[i] = in48[47 - i];
  xle[47] &= 0x1f;
  uint64_t xw[6];
  for (int i = 0; i < 6; ++i) { uint64_t w = 0; for (int j = 7; j >= 0; --j) w = (w << 8) | xle[8 * i + j]; xw[i] = w; }
  if (ge_p(xw)) return 1;
  fq x = fq_from_le(xle);
  uint8_t four[48]; memset(four, 0, sizeof four); four[0] = 4;
  fq rhs = fq_add(fq_mul(fq_mul(x, x), x), fq_from_le(four));
  uint8_t one[48]; memset(one, 0, sizeof one); one[0] = 1;
  fq y = fq_from_le(one);
  for (int i = 383; i >= 0; --i) { y = fq_mul(y, y); if ((SQRT_EXP[i >> 6] >> (i & 63)) & 1) y = fq_mul(y, rhs); }
  if (!fq_equal(fq_mul(y, y), rhs)) return 2;
  uint8_t a96[96], enc[48];
  memcpy(a96, xle, 48);
  fq_to_le(y, a96 + 48);
  orc_compress(a96, enc);
  if (((enc[0] & 0x20) != 0) != ((flags & 0x20) != 0)) {
    fq zero; memset(&zero, 0, sizeof zero);
    fq_to_le(fq_sub(zero, y), a96 + 48);
  }
  if (check_subgroup) {
    pt base = pt_from_affine96(a96), acc = pt_identity();
    for (int bit = 254; bit >= 0; --bit) {
      acc = pt_double(acc);
      if ((GROUP_ORDER[bit >> 6] >> (bit & 63)) & 1) acc = pt_add(acc, base);
    }
    if (!fq_is_zero(acc.z)) return 3;
  }
  memcpy(out96, a96, 96);
  return 0;
}
