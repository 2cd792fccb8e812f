// TEST-ONLY: host-side C++ of the product (host_g1.cpp, merlin.cpp) under AddressSanitizer + UBSan.
// (GPU sanitizers are not available on the pool; the device arithmetic is range-checked by fp28_harness.cpp.)
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../curdleproofs_pie_amd/csrc/host_g1.h"

extern "C" {
void cg1_merlin_init(uint8_t* state, const uint8_t* label, size_t len);
void cg1_merlin_append(uint8_t* state, const uint8_t* label, size_t llen, const uint8_t* msg, size_t mlen);
void cg1_merlin_append_list(uint8_t* state, const uint8_t* label, size_t llen, const uint8_t* items, size_t item_len, size_t count);
void cg1_merlin_challenge(uint8_t* state, const uint8_t* label, size_t llen, uint8_t* out, size_t n);
void cg1_merlin_challenge_scalar(uint8_t* state, const uint8_t* label, size_t llen, uint8_t out32[32]);
}

using namespace cg1h;

static uint64_t rng_state = 0x1234567;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main() {
  jac g = jac_generator();
  std::vector<jac> pts;
  for (int i = 0; i < 40; ++i) {
    uint8_t k[32];
    for (int j = 0; j < 32; ++j) k[j] = (uint8_t)rnd();
    k[31] &= 0x3f;
    pts.push_back(jac_mul(g, k));
  }
  pts.push_back(jac_identity());
  // group law identities
  for (size_t i = 0; i + 1 < pts.size(); ++i) {
    jac s = jac_add(pts[i], pts[i + 1]);
    jac d = jac_add(s, jac_neg(pts[i + 1]));
    if (!jac_eq(d, pts[i])) { printf("add/sub mismatch\n"); return 1; }
    if (!jac_eq(jac_add(pts[i], pts[i]), jac_dbl(pts[i]))) { printf("dbl mismatch\n"); return 1; }
    if (!jac_on_curve(s)) { printf("off curve\n"); return 1; }
  }
  // codec round trips, batch normalisation
  std::vector<fe> xs(pts.size()), ys(pts.size());
  std::vector<uint8_t> inf(pts.size());
  jac_batch_to_affine(pts.data(), pts.size(), xs.data(), ys.data(), inf.data());
  for (size_t i = 0; i < pts.size(); ++i) {
    uint8_t c[48];
    g1_compress(pts[i], c);
    jac back;
    if (g1_decompress(c, true, back) != 0 || !jac_eq(back, pts[i])) { printf("codec mismatch %zu\n", i); return 1; }
    if (!inf[i] && !jac_eq(jac_from_affine(xs[i], ys[i]), pts[i])) { printf("batch affine mismatch\n"); return 1; }
  }
  uint8_t bad[48];
  memset(bad, 0xff, sizeof bad);
  jac tmp;
  // the infinity flag decides alone (the wheel's decoder, host_g1.cpp): all-ones is the identity, not an error
  if (g1_decompress(bad, false, tmp) != 0 || !jac_is_identity(tmp)) { printf("infinity flag not honoured\n"); return 1; }
  bad[0] = 0x9f;                       // compressed, finite, x = 2^381 - 1 >= p
  if (g1_decompress(bad, false, tmp) == 0) { printf("accepted garbage\n"); return 1; }
  if (!jac_in_subgroup(pts[3])) { printf("subgroup\n"); return 1; }
  // transcript
  uint8_t st[208], out[400], sc[32];
  cg1_merlin_init(st, (const uint8_t*)"sanitize", 8);
  std::vector<uint8_t> msg(5000);
  for (auto& b : msg) b = (uint8_t)rnd();
  for (size_t len : {0u, 1u, 165u, 166u, 167u, 4999u}) cg1_merlin_append(st, (const uint8_t*)"m", 1, msg.data(), len);
  cg1_merlin_append_list(st, (const uint8_t*)"list", 4, msg.data(), 48, 100);
  cg1_merlin_challenge(st, (const uint8_t*)"c", 1, out, 400);
  cg1_merlin_challenge(st, (const uint8_t*)"c", 1, out, 0);
  cg1_merlin_challenge_scalar(st, (const uint8_t*)"s", 1, sc);
  // the dedicated squaring (fe_mul_x86.h fe_sqr_adx) against the product, random and edge operands below p
  {
    static const uint64_t P[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull, 0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
    auto same = [](const fe& a) { fe s = fe_sqr(a), m = fe_mul(a, a); return memcmp(s.l, m.l, 48) == 0; };
    fe a;
    for (int i = 0; i < 200000; ++i) {
      for (int j = 0; j < 6; ++j) a.l[j] = (i % 5 == 0 && (rnd() & 1)) ? ((rnd() & 1) ? ~0ull : 0ull) : rnd();
      a.l[5] %= P[5];
      if (!same(a)) { printf("fe_sqr differs from fe_mul at case %d\n", i); return 1; }
    }
    memset(a.l, 0, 48); if (!same(a)) return 1;
    memcpy(a.l, P, 48); a.l[0] -= 1; if (!same(a)) return 1;
    for (int k = 0; k < 380; ++k) {
      memset(a.l, 0, 48); a.l[k / 64] = 1ull << (k % 64); if (!same(a)) return 1;
      a.l[k / 64] -= 1; for (int j = 0; j < k / 64; ++j) a.l[j] = ~0ull; if (!same(a)) return 1;
    }
  }
  printf("sanitize ok %02x%02x\n", out[0], sc[0]);
  return 0;
}
