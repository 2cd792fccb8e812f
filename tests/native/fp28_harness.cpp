// TEST-ONLY harness: compiles the *device* field/group headers (fp28.h, g1_xyzz.h) for the host with
// CG1_CHECK_BOUNDS so every 64-bit column accumulator, lazy add and lazy subtract is range-checked
// (abort on violation).  Driven from tests/test_fp28_host.py through ctypes and compared against the
// big-int oracle.  Never linked into the product library.
#include <cstring>
#include "../../curdleproofs_pie_amd/csrc/g1_xyzz.h"

using namespace cg1;

static fp load_mont(const uint8_t* le48) {
  uint32_t w[12];
  memcpy(w, le48, 48);
  return fp_to_mont(fp_from_words(w));
}
static void store_std(const fp& a, uint8_t* le48) {
  uint32_t w[12];
  fp_to_words(a, w);
  memcpy(le48, w, 48);
}

extern "C" {

void t_fp_mul(const uint8_t* a, const uint8_t* b, uint8_t* out) { store_std(fp_mul(load_mont(a), load_mont(b)), out); }
void t_fp_sqr(const uint8_t* a, uint8_t* out) { store_std(fp_sqr(load_mont(a)), out); }
void t_fp_inv(const uint8_t* a, uint8_t* out) { store_std(fp_inv(load_mont(a)), out); }
// (a - b) * c and (a + b) * c : exercises the lazy forms feeding a product
void t_fp_submul(const uint8_t* a, const uint8_t* b, const uint8_t* c, uint8_t* out) {
  store_std(fp_mul(fp_sub<3>(load_mont(a), load_mont(b)), load_mont(c)), out);
}
void t_fp_addmul(const uint8_t* a, const uint8_t* b, const uint8_t* c, uint8_t* out) {
  store_std(fp_mul(fp_add(load_mont(a), load_mont(b)), load_mont(c)), out);
}
int t_fp_is_zero_diff(const uint8_t* a, const uint8_t* b) {   // a - b + 12p == 0 mod p ?
  return fp_is_zero_mod_p(fp_sub<12>(load_mont(a), load_mont(b)), 14) ? 1 : 0;
}

// Worst-case limb magnitudes admitted by the static bound analysis of g1_xyzz.h: every 64-bit column
// accumulator must still fit (mad64 asserts).  Values are meaningless; only magnitudes matter.
int t_worst_case_bounds() {
  fp lazy, neg, norm;
  for (int i = 0; i < NL; ++i) { lazy.l[i] = 0x2FFFFFFFu; neg.l[i] = 0x1FFFFFFFu; norm.l[i] = 0x0FFFFFFFu; }
  lazy.l[NL - 1] = 0x00FFFFFFu; neg.l[NL - 1] = 0x00FFFFFFu; norm.l[NL - 1] = 0x000FFFFFu;   // top limbs are small
  fp r1 = fp_mul(lazy, lazy);            // (sub result) x (sub result): R*(Q-X3), M*(S-X3)
  fp r2 = fp_sqr(lazy);                  // P^2, R^2, M^2
  fp r3 = fp_mul2(lazy, lazy, norm, neg);  // fused Y3
  fp r4 = fp_mul(neg, norm);             // lazily negated y times ZZZ
  // jacp_dbl: E = 3A times (D - X3); 2 Y1 (Y1 in its lazy post-doubling form) times Z1; X1 times 4B
  fp e3, y2, b4;
  for (int i = 0; i < NL; ++i) { e3.l[i] = 3u * norm.l[i]; y2.l[i] = 2u * lazy.l[i]; b4.l[i] = 4u * norm.l[i]; }
  fp r5 = fp_mul(e3, lazy), r6 = fp_mul(y2, norm), r7 = fp_mul(norm, b4), r8 = fp_sqr(e3);
  return (int)((r1.l[0] ^ r2.l[0] ^ r3.l[0] ^ r4.l[0] ^ r5.l[0] ^ r6.l[0] ^ r7.l[0] ^ r8.l[0]) & 1u) | 2;
}

static void export_xyzz(const xyzz& a, uint8_t* out /* 4*48 + 4 */) {
  xyzz_words o;
  xyzz_export(a, o);
  memcpy(out, o.w, 4 * 48);
  memcpy(out + 4 * 48, &o.inf, 4);
}

// acc = sum_i (+/-) P_i using the mixed add, points given as affine 96-byte records (x||y LE, std form).
// neg[i] != 0 negates y lazily exactly as the bucket kernel does.
void t_madd_seq(const uint8_t* pts96, const uint8_t* neg, int n, uint8_t* out) {
  xyzz acc = xyzz_identity();
  for (int i = 0; i < n; ++i) {
    fp x = load_mont(pts96 + 96 * i), y = load_mont(pts96 + 96 * i + 48);
    if (neg[i]) y = fp_neg<3>(y);
    acc = xyzz_madd(acc, x, y);
  }
  export_xyzz(acc, out);
}

// The same sum the way k_accumulate forms a chunk: the first two entries through the affine+affine addition, the rest
// through the mixed add (its in-place common case first, as the kernel's hot loop does).
void t_chunk_seq(const uint8_t* pts96, const uint8_t* neg, int n, uint8_t* out) {
  xyzz acc = xyzz_identity();
  auto ld = [&](int i, fp& x, fp& y) {
    x = load_mont(pts96 + 96 * i); y = load_mont(pts96 + 96 * i + 48);
    if (neg[i]) y = fp_neg<3>(y);
  };
  fp x, y;
  if (n >= 2) {
    fp x1, y1;
    ld(0, x, y); ld(1, x1, y1);
    acc = xyzz_mmadd(x, y, x1, y1);
  } else if (n == 1) {
    ld(0, x, y);
    acc = xyzz_from_affine(x, y);
  }
  for (int i = 2; i < n; ++i) {
    ld(i, x, y);
    if (!xyzz_madd_fast(acc, x, y)) acc = xyzz_madd(acc, x, y);     // the kernel's hot loop, then its exceptional-case tail
  }
  export_xyzz(acc, out);
}

// Sum the same sequence as a balanced tree of full adds (exercises xyzz_add, incl. P+P and P-P).
void t_add_tree(const uint8_t* pts96, const uint8_t* neg, int n, uint8_t* out) {
  if (n == 0) { export_xyzz(xyzz_identity(), out); return; }
  xyzz* v = new xyzz[n];
  for (int i = 0; i < n; ++i) {
    fp x = load_mont(pts96 + 96 * i), y = load_mont(pts96 + 96 * i + 48);
    if (neg[i]) y = fp_neg<3>(y);
    v[i] = xyzz_from_affine(x, y);
  }
  for (int m = n; m > 1; m = (m + 1) / 2)
    for (int i = 0; i < m / 2; ++i) v[i] = xyzz_add(v[i], v[m - 1 - i]);
  export_xyzz(v[0], out);
  delete[] v;
}

// k * P by double-and-add over XYZZ (dbl + madd), k given as 32 LE bytes.
void t_scalar_mul(const uint8_t* pt96, const uint8_t* k32, uint8_t* out) {
  fp x = load_mont(pt96), y = load_mont(pt96 + 48);
  xyzz acc = xyzz_identity();
  for (int bit = 255; bit >= 0; --bit) {
    acc = xyzz_dbl(acc);
    if ((k32[bit >> 3] >> (bit & 7)) & 1) acc = xyzz_madd(acc, x, y);
  }
  export_xyzz(acc, out);
}

// running-sum pattern of the bucket reduction: run += B_k; tot += run  (k from high to low)
void t_running_sum(const uint8_t* pts96, int n, uint8_t* out) {
  xyzz run = xyzz_identity(), tot = xyzz_identity();
  for (int i = n - 1; i >= 0; --i) {
    fp x = load_mont(pts96 + 96 * i), y = load_mont(pts96 + 96 * i + 48);
    run = xyzz_add(run, xyzz_from_affine(x, y));
    tot = xyzz_add(tot, run);
  }
  export_xyzz(tot, out);
}

// ---- Jacobian coordinates (the subgroup test's doublings): k * P by double-and-add (jacp_dbl + jacp_madd), exported through XYZZ
static void export_jacp(const jacp& a, uint8_t* out) {
  xyzz t;
  t.inf = a.inf;
  if (!a.inf) { t.X = a.X; t.Y = fp_mul(a.Y, fp_one()); t.ZZ = fp_sqr(a.Z); t.ZZZ = fp_mul(t.ZZ, a.Z); }
  else { t.X = t.Y = t.ZZ = t.ZZZ = fp_zero(); }
  export_xyzz(t, out);
}
void t_scalar_mul_jac(const uint8_t* pt96, const uint8_t* k32, uint8_t* out) {
  fp x = load_mont(pt96), y = load_mont(pt96 + 48);
  jacp acc = jacp_identity();
  for (int bit = 255; bit >= 0; --bit) {
    acc = jacp_dbl(acc);
    if ((k32[bit >> 3] >> (bit & 7)) & 1) acc = jacp_madd(acc, x, y);
  }
  export_jacp(acc, out);
}
// sum of n affine points as a balanced tree of jacp_add (incl. P + P, P - P, identity operands); neg[i] negates y lazily
void t_add_tree_jac(const uint8_t* pts96, const uint8_t* neg, int n, uint8_t* out) {
  jacp* v = new jacp[n + 1];
  for (int i = 0; i < n; ++i) {
    fp x = load_mont(pts96 + 96 * i), y = load_mont(pts96 + 96 * i + 48);
    v[i] = jacp_dbl(jacp_from_affine(x, neg[i] ? fp_neg<3>(y) : y));      // doubled first: Z != 1 and Y in its lazy post-doubling form
  }
  int m = n;
  if (m == 0) { export_jacp(jacp_identity(), out); delete[] v; return; }
  while (m > 1) {
    for (int i = 0; i < m / 2; ++i) v[i] = jacp_add(v[i], v[m - 1 - i]);
    m = (m + 1) / 2;
  }
  export_jacp(v[0], out);
  delete[] v;
}
// the checked decompression's subgroup test on a curve point given as affine96 (x || y, standard form)
int t_in_subgroup(const uint8_t* pt96) { return g1_in_subgroup(load_mont(pt96), load_mont(pt96 + 48)) ? 1 : 0; }

}  // extern "C"
