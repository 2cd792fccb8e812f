// ASan/UBSan driver for the shuffle-verifier front-end (csrc/shuffle_verify.cpp): a golden proof, bit-flipped
// copies of it and random garbage through cg1_shuffle_prepare, single- and multi-threaded.  Input file (written by
// tests/test_sanitizers.py from tests/golden/shuffle_vectors.json): u64 ell | crs bytes | instance | proof.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/curdle_g1.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  uint64_t ell = 0;
  if (fread(&ell, 8, 1, f) != 1) return 2;
  std::vector<uint8_t> crsb((ell + 9) * 48), inst(4 * ell * 48);
  if (fread(crsb.data(), 1, crsb.size(), f) != crsb.size() || fread(inst.data(), 1, inst.size(), f) != inst.size()) return 2;
  cg1_shuffle_crs* crs = cg1_shuffle_crs_create(crsb.data(), ell, 4);
  if (!crs) { printf("crs rejected\n"); return 1; }
  if (cg1_shuffle_crs_create(crsb.data(), ell + 1, 4) != nullptr) { printf("bad ell accepted\n"); return 1; }
  const size_t pb = cg1_shuffle_proof_bytes(crs), L = cg1_shuffle_points_per_proof(crs), C = cg1_shuffle_crs_points(crs),
               nch = cg1_shuffle_challenges_per_proof(crs);
  std::vector<uint8_t> proof(pb);
  if (fread(proof.data(), 1, pb, f) != pb) return 2;
  fclose(f);

  const size_t n = 24;
  std::vector<uint8_t> insts, proofs, weights(n * 12 * 32);
  for (size_t i = 0; i < n; ++i) {
    std::vector<uint8_t> p = proof, in = inst;
    if (i >= 1 && i < 12) p[rnd() % pb] ^= (uint8_t)(1u << (rnd() % 8));          // single bit flips
    if (i >= 12 && i < 16) in[rnd() % in.size()] ^= (uint8_t)(1u << (rnd() % 8));
    if (i >= 16 && i < 20) for (auto& b : p) b = (uint8_t)rnd();                  // garbage proof
    if (i >= 20) for (auto& b : in) b = (uint8_t)rnd();                           // garbage instance
    insts.insert(insts.end(), in.begin(), in.end());
    proofs.insert(proofs.end(), p.begin(), p.end());
  }
  for (size_t i = 0; i < weights.size(); ++i) weights[i] = (uint8_t)rnd();
  for (size_t i = 31; i < weights.size(); i += 32) weights[i] &= 0x3f;
  std::vector<uint8_t> pts(n * L * 48), sc(n * L * 32), cs(n * C * 32), ch(n * nch * 32), pts2(n * L * 48), sc2(n * L * 32), cs2(n * C * 32);
  std::vector<int32_t> st(n), st2(n);
  if (cg1_shuffle_prepare(crs, n, insts.data(), proofs.data(), weights.data(), nullptr, 0, pts.data(), sc.data(), cs.data(), st.data(), ch.data(), 1)) return 1;
  if (cg1_shuffle_prepare(crs, n, insts.data(), proofs.data(), weights.data(), nullptr, 0, pts2.data(), sc2.data(), cs2.data(), st2.data(), nullptr, 4)) return 1;
  if (st[0] != 0) { printf("golden proof rejected by the front-end (%d)\n", st[0]); return 1; }
  if (st != st2 || sc != sc2 || cs != cs2 || pts != pts2) { printf("thread counts disagree\n"); return 1; }
  std::vector<uint8_t> wire(n * L * 48);
  if (cg1_shuffle_gather_points(crs, n, insts.data(), proofs.data(), wire.data())) { printf("gather failed\n"); return 1; }
  // the front-end's copy carries what is HASHED: an encoding with the infinity flag set is replaced by the canonical
  // 0xC0 00 .. 00 (the reference hashes re-serialised points); everything else is the wire bytes
  for (size_t i = 0; i < n * L; ++i) {
    const uint8_t* w = wire.data() + 48 * i;
    const uint8_t* q = pts.data() + 48 * i;
    bool same = memcmp(w, q, 48) == 0;
    if (!same && (w[0] & 0xC0) == 0xC0) {
      same = q[0] == 0xC0;
      for (int k = 1; k < 48; ++k) same = same && q[k] == 0;
    }
    if (!same) { printf("gather mismatch at point %zu\n", i); return 1; }
  }
  std::vector<uint8_t> pstat(n * L, 0), sum(C * 32);
  pstat[3 * L + 5] = 3;
  if (cg1_shuffle_apply_point_status(st.data(), pstat.data(), n, L, sc.data(), cs.data(), C)) return 1;
  if (st[3] == 0) { printf("point status not applied\n"); return 1; }
  if (cg1_shuffle_sum_crs_scalars(cs.data(), st.data(), n, C, sum.data())) return 1;
  size_t rejected = 0;
  for (size_t i = 0; i < n; ++i) rejected += st[i] != 0;
  printf("front-end rejected %zu of %zu\n", rejected, n);
  {   // opening-proof front-end: 700 proofs of random bytes (pool path) -- statuses only, no crash / race
    const size_t m = 700;
    std::vector<uint8_t> trk(m * 96), kc(m * 48), pf(m * 128), w(m * 64), op(m * 240), os(m * 160), og(m * 32);
    std::vector<int32_t> ost(m);
    for (auto& b : trk) b = (uint8_t)rnd();
    for (auto& b : kc) b = (uint8_t)rnd();
    for (auto& b : pf) b = (uint8_t)rnd();
    for (auto& b : w) b = (uint8_t)rnd();
    for (size_t i = 31; i < w.size(); i += 32) w[i] &= 0x3f;
    for (size_t i = 127; i < pf.size(); i += 256) pf[i] &= 0x3f;           // every other proof gets a canonical s
    if (cg1_opening_prepare(m, trk.data(), kc.data(), pf.data(), w.data(), op.data(), os.data(), og.data(), ost.data())) return 1;
    size_t okc = 0;
    for (size_t i = 0; i < m; ++i) okc += ost[i] == 0;
    printf("opening front-end prepared %zu of %zu\n", okc, m);
  }
  cg1_shuffle_crs_destroy(crs);
  printf("sanitize ok\n");
  return 0;
}
