// Native Merlin v1.0 transcript (STROBE-128 over Keccak-f[1600]) -- SURVEY.md 8(f) row 1.
//
// Stands behind /root/reference/merlin_transcripts/merlin_transcripts/{merlin_transcript.py:6-24,
// strobe.py:16-107, keccak.py:16-66} and the Fiat-Shamir adaptor curdleproofs/curdleproofs/
// curdleproofs_transcript.py:7-28.  The reference's pure-Python Keccak costs ~0.8 ms per permutation and a
// N=128 shuffle verification makes ~727 of them (~0.57 s of hashing per proof); this is the same
// construction in C++ (~1 us per permutation).  Host code by design: one transcript is a strictly serial
// sponge; the batched (one transcript per lane) HIP version is a later row.
//
// State is a caller-owned 208-byte blob (200-byte sponge, pos, pos_begin, cur_flags): no handles, no globals.
#include <cstdint>
#include <cstring>
#include "../../include/curdle_g1.h"

namespace {

constexpr int STROBE_R = 166;                       // strobe.py:4
constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32;   // strobe.py:6-11

struct Strobe {
  uint8_t st[200];
  uint8_t pos, pos_begin, cur_flags, pad[5];
};
static_assert(sizeof(Strobe) == CG1_MERLIN_STATE_BYTES, "merlin state blob size");

inline uint64_t rotl(uint64_t v, int n) { return (v << n) | (v >> (64 - n)); }

// Theta, rho+pi and chi with every lane index a compile-time constant (fully unrolled round; the 25 lanes live in
// registers), two rounds per loop trip.  The host is little-endian, so lanes load/store with memcpy.
#define CG1_KECCAK_ROUND(A, E, rc)                                                                  \
  {                                                                                                 \
    const uint64_t c0 = A[0] ^ A[5] ^ A[10] ^ A[15] ^ A[20], c1 = A[1] ^ A[6] ^ A[11] ^ A[16] ^ A[21],  \
                   c2 = A[2] ^ A[7] ^ A[12] ^ A[17] ^ A[22], c3 = A[3] ^ A[8] ^ A[13] ^ A[18] ^ A[23],  \
                   c4 = A[4] ^ A[9] ^ A[14] ^ A[19] ^ A[24];                                        \
    const uint64_t d0 = c4 ^ rotl(c1, 1), d1 = c0 ^ rotl(c2, 1), d2 = c1 ^ rotl(c3, 1), d3 = c2 ^ rotl(c4, 1), \
                   d4 = c3 ^ rotl(c0, 1);                                                           \
    uint64_t b0, b1, b2, b3, b4;                                                                    \
    b0 = A[0] ^ d0; b1 = rotl(A[6] ^ d1, 44); b2 = rotl(A[12] ^ d2, 43); b3 = rotl(A[18] ^ d3, 21); b4 = rotl(A[24] ^ d4, 14); \
    E[0] = b0 ^ (~b1 & b2) ^ (rc); E[1] = b1 ^ (~b2 & b3); E[2] = b2 ^ (~b3 & b4); E[3] = b3 ^ (~b4 & b0); E[4] = b4 ^ (~b0 & b1); \
    b0 = rotl(A[3] ^ d3, 28); b1 = rotl(A[9] ^ d4, 20); b2 = rotl(A[10] ^ d0, 3); b3 = rotl(A[16] ^ d1, 45); b4 = rotl(A[22] ^ d2, 61); \
    E[5] = b0 ^ (~b1 & b2); E[6] = b1 ^ (~b2 & b3); E[7] = b2 ^ (~b3 & b4); E[8] = b3 ^ (~b4 & b0); E[9] = b4 ^ (~b0 & b1); \
    b0 = rotl(A[1] ^ d1, 1); b1 = rotl(A[7] ^ d2, 6); b2 = rotl(A[13] ^ d3, 25); b3 = rotl(A[19] ^ d4, 8); b4 = rotl(A[20] ^ d0, 18); \
    E[10] = b0 ^ (~b1 & b2); E[11] = b1 ^ (~b2 & b3); E[12] = b2 ^ (~b3 & b4); E[13] = b3 ^ (~b4 & b0); E[14] = b4 ^ (~b0 & b1); \
    b0 = rotl(A[4] ^ d4, 27); b1 = rotl(A[5] ^ d0, 36); b2 = rotl(A[11] ^ d1, 10); b3 = rotl(A[17] ^ d2, 15); b4 = rotl(A[23] ^ d3, 56); \
    E[15] = b0 ^ (~b1 & b2); E[16] = b1 ^ (~b2 & b3); E[17] = b2 ^ (~b3 & b4); E[18] = b3 ^ (~b4 & b0); E[19] = b4 ^ (~b0 & b1); \
    b0 = rotl(A[2] ^ d2, 62); b1 = rotl(A[8] ^ d3, 55); b2 = rotl(A[14] ^ d4, 39); b3 = rotl(A[15] ^ d0, 41); b4 = rotl(A[21] ^ d1, 2); \
    E[20] = b0 ^ (~b1 & b2); E[21] = b1 ^ (~b2 & b3); E[22] = b2 ^ (~b3 & b4); E[23] = b3 ^ (~b4 & b0); E[24] = b4 ^ (~b0 & b1); \
  }

void keccak_f1600(uint8_t* bytes) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
      0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
      0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
      0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  uint64_t a[25], e[25];
  memcpy(a, bytes, 200);                    // lanes are little-endian 64-bit words, lane index x + 5y
  for (int round = 0; round < 24; round += 2) {
    CG1_KECCAK_ROUND(a, e, RC[round]);
    CG1_KECCAK_ROUND(e, a, RC[round + 1]);
  }
  memcpy(bytes, a, 200);
}
#undef CG1_KECCAK_ROUND

// ---- eight permutations at once: lane w of state k at a[w][k] (SoA), one 512-bit vector per lane index.
// Written with compiler vector extensions; the same body is compiled twice -- for AVX-512 (vprolq / vpternlogq,
// selected at run time) and for the baseline ISA.  Used by the batch front-end, which advances many independent
// transcripts in step (csrc/shuffle_verify.cpp).
typedef uint64_t v8u64 __attribute__((vector_size(64), aligned(8)));
#define vrotl(v, n) (((v) << (n)) | ((v) >> (64 - (n))))      // a macro: vector arguments must not cross a target("avx512f") boundary

#define CG1_KECCAK_ROUND_V(A, E, rc)                                                                \
  {                                                                                                 \
    const v8u64 c0 = A[0] ^ A[5] ^ A[10] ^ A[15] ^ A[20], c1 = A[1] ^ A[6] ^ A[11] ^ A[16] ^ A[21],    \
                c2 = A[2] ^ A[7] ^ A[12] ^ A[17] ^ A[22], c3 = A[3] ^ A[8] ^ A[13] ^ A[18] ^ A[23],    \
                c4 = A[4] ^ A[9] ^ A[14] ^ A[19] ^ A[24];                                           \
    const v8u64 d0 = c4 ^ vrotl(c1, 1), d1 = c0 ^ vrotl(c2, 1), d2 = c1 ^ vrotl(c3, 1), d3 = c2 ^ vrotl(c4, 1), \
                d4 = c3 ^ vrotl(c0, 1);                                                             \
    v8u64 b0, b1, b2, b3, b4;                                                                       \
    b0 = A[0] ^ d0; b1 = vrotl(A[6] ^ d1, 44); b2 = vrotl(A[12] ^ d2, 43); b3 = vrotl(A[18] ^ d3, 21); b4 = vrotl(A[24] ^ d4, 14); \
    E[0] = b0 ^ (~b1 & b2) ^ (rc); E[1] = b1 ^ (~b2 & b3); E[2] = b2 ^ (~b3 & b4); E[3] = b3 ^ (~b4 & b0); E[4] = b4 ^ (~b0 & b1); \
    b0 = vrotl(A[3] ^ d3, 28); b1 = vrotl(A[9] ^ d4, 20); b2 = vrotl(A[10] ^ d0, 3); b3 = vrotl(A[16] ^ d1, 45); b4 = vrotl(A[22] ^ d2, 61); \
    E[5] = b0 ^ (~b1 & b2); E[6] = b1 ^ (~b2 & b3); E[7] = b2 ^ (~b3 & b4); E[8] = b3 ^ (~b4 & b0); E[9] = b4 ^ (~b0 & b1); \
    b0 = vrotl(A[1] ^ d1, 1); b1 = vrotl(A[7] ^ d2, 6); b2 = vrotl(A[13] ^ d3, 25); b3 = vrotl(A[19] ^ d4, 8); b4 = vrotl(A[20] ^ d0, 18); \
    E[10] = b0 ^ (~b1 & b2); E[11] = b1 ^ (~b2 & b3); E[12] = b2 ^ (~b3 & b4); E[13] = b3 ^ (~b4 & b0); E[14] = b4 ^ (~b0 & b1); \
    b0 = vrotl(A[4] ^ d4, 27); b1 = vrotl(A[5] ^ d0, 36); b2 = vrotl(A[11] ^ d1, 10); b3 = vrotl(A[17] ^ d2, 15); b4 = vrotl(A[23] ^ d3, 56); \
    E[15] = b0 ^ (~b1 & b2); E[16] = b1 ^ (~b2 & b3); E[17] = b2 ^ (~b3 & b4); E[18] = b3 ^ (~b4 & b0); E[19] = b4 ^ (~b0 & b1); \
    b0 = vrotl(A[2] ^ d2, 62); b1 = vrotl(A[8] ^ d3, 55); b2 = vrotl(A[14] ^ d4, 39); b3 = vrotl(A[15] ^ d0, 41); b4 = vrotl(A[21] ^ d1, 2); \
    E[20] = b0 ^ (~b1 & b2); E[21] = b1 ^ (~b2 & b3); E[22] = b2 ^ (~b3 & b4); E[23] = b3 ^ (~b4 & b0); E[24] = b4 ^ (~b0 & b1); \
  }

static inline __attribute__((always_inline)) void keccak_x8_body(uint64_t* lanes) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
      0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
      0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
      0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  v8u64 a[25], e[25];
  for (int i = 0; i < 25; ++i) memcpy(&a[i], lanes + 8 * i, 64);
  for (int round = 0; round < 24; round += 2) {
    const v8u64 r0 = {RC[round], RC[round], RC[round], RC[round], RC[round], RC[round], RC[round], RC[round]};
    const uint64_t q = RC[round + 1];
    const v8u64 r1 = {q, q, q, q, q, q, q, q};
    CG1_KECCAK_ROUND_V(a, e, r0);
    CG1_KECCAK_ROUND_V(e, a, r1);
  }
  for (int i = 0; i < 25; ++i) memcpy(lanes + 8 * i, &a[i], 64);
}
__attribute__((target("avx512f,avx512vl,avx512dq,avx512bw"))) void keccak_x8_avx512(uint64_t* lanes) { keccak_x8_body(lanes); }
void keccak_x8_generic(uint64_t* lanes) { keccak_x8_body(lanes); }

}  // namespace
#include <immintrin.h>
namespace {

// 8 x 8 transpose of 64-bit elements: rows r[0..7] -> columns (an involution: the same routine goes both ways)
__attribute__((target("avx512f"))) static inline void transpose8x8(__m512i r[8]) {
  const __m512i t0 = _mm512_unpacklo_epi64(r[0], r[1]), t1 = _mm512_unpackhi_epi64(r[0], r[1]);
  const __m512i t2 = _mm512_unpacklo_epi64(r[2], r[3]), t3 = _mm512_unpackhi_epi64(r[2], r[3]);
  const __m512i t4 = _mm512_unpacklo_epi64(r[4], r[5]), t5 = _mm512_unpackhi_epi64(r[4], r[5]);
  const __m512i t6 = _mm512_unpacklo_epi64(r[6], r[7]), t7 = _mm512_unpackhi_epi64(r[6], r[7]);
  const __m512i u0 = _mm512_shuffle_i64x2(t0, t2, 0x88), u1 = _mm512_shuffle_i64x2(t0, t2, 0xdd);
  const __m512i u2 = _mm512_shuffle_i64x2(t4, t6, 0x88), u3 = _mm512_shuffle_i64x2(t4, t6, 0xdd);
  const __m512i u4 = _mm512_shuffle_i64x2(t1, t3, 0x88), u5 = _mm512_shuffle_i64x2(t1, t3, 0xdd);
  const __m512i u6 = _mm512_shuffle_i64x2(t5, t7, 0x88), u7 = _mm512_shuffle_i64x2(t5, t7, 0xdd);
  r[0] = _mm512_shuffle_i64x2(u0, u2, 0x88); r[4] = _mm512_shuffle_i64x2(u0, u2, 0xdd);
  r[2] = _mm512_shuffle_i64x2(u1, u3, 0x88); r[6] = _mm512_shuffle_i64x2(u1, u3, 0xdd);
  r[1] = _mm512_shuffle_i64x2(u4, u6, 0x88); r[5] = _mm512_shuffle_i64x2(u4, u6, 0xdd);
  r[3] = _mm512_shuffle_i64x2(u5, u7, 0x88); r[7] = _mm512_shuffle_i64x2(u5, u7, 0xdd);
}

// Eight 200-byte sponges where they lie (array-of-states): load + transpose in registers, permute, transpose back.
// Avoids the 2 x 200 scalar moves per call of a staging buffer (and their failed store-to-load forwarding).
__attribute__((target("avx512f,avx512vl,avx512dq,avx512bw"))) void keccak_x8_aos_avx512(uint8_t* const st[8], int live) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
      0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
      0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
      0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  v8u64 a[25], e[25];
  for (int b = 0; b < 3; ++b) {
    __m512i r[8];
    for (int j = 0; j < 8; ++j) r[j] = _mm512_loadu_si512(st[j] + 64 * b);
    transpose8x8(r);
    for (int j = 0; j < 8; ++j) a[8 * b + j] = (v8u64)r[j];
  }
  {
    uint64_t last[8];
    for (int j = 0; j < 8; ++j) memcpy(&last[j], st[j] + 192, 8);
    memcpy(&a[24], last, 64);
  }
  for (int round = 0; round < 24; round += 2) {
    const v8u64 r0 = {RC[round], RC[round], RC[round], RC[round], RC[round], RC[round], RC[round], RC[round]};
    const uint64_t q = RC[round + 1];
    const v8u64 r1 = {q, q, q, q, q, q, q, q};
    CG1_KECCAK_ROUND_V(a, e, r0);
    CG1_KECCAK_ROUND_V(e, a, r1);
  }
  for (int b = 0; b < 3; ++b) {
    __m512i r[8];
    for (int j = 0; j < 8; ++j) r[j] = (__m512i)a[8 * b + j];
    transpose8x8(r);
    for (int j = 0; j < live; ++j) _mm512_storeu_si512(st[j] + 64 * b, r[j]);
  }
  {
    uint64_t last[8];
    memcpy(last, &a[24], 64);
    for (int j = 0; j < live; ++j) memcpy(st[j] + 192, &last[j], 8);
  }
}
#undef CG1_KECCAK_ROUND_V

void run_f(Strobe& s) {                              // strobe.py:55-61
  s.st[s.pos] ^= s.pos_begin;
  s.st[s.pos + 1] ^= 0x04;
  s.st[STROBE_R + 1] ^= 0x80;
  keccak_f1600(s.st);
  s.pos = 0; s.pos_begin = 0;
}
void absorb(Strobe& s, const uint8_t* d, size_t n) {  // strobe.py:63-68
  while (n) {
    size_t room = (size_t)STROBE_R - s.pos, k = n < room ? n : room;
    uint8_t* dst = s.st + s.pos;
    for (size_t i = 0; i < k; ++i) dst[i] ^= d[i];
    s.pos = (uint8_t)(s.pos + k);
    d += k;
    n -= k;
    if (s.pos == STROBE_R) run_f(s);
  }
}
void overwrite(Strobe& s, const uint8_t* d, size_t n) {   // strobe.py:70-75
  for (size_t i = 0; i < n; ++i) { s.st[s.pos++] = d[i]; if (s.pos == STROBE_R) run_f(s); }
}
void squeeze(Strobe& s, uint8_t* out, size_t n) {     // strobe.py:77-87
  for (size_t i = 0; i < n; ++i) { out[i] = s.st[s.pos]; s.st[s.pos++] = 0; if (s.pos == STROBE_R) run_f(s); }
}
int begin_op(Strobe& s, uint8_t flags, bool more) {   // strobe.py:89-107
  if (more) return s.cur_flags == flags ? 0 : CG1_ERR_ARG;
  if (flags & FLAG_T) return CG1_ERR_ARG;
  uint8_t hdr[2] = {s.pos_begin, flags};
  s.pos_begin = (uint8_t)(s.pos + 1);
  s.cur_flags = flags;
  absorb(s, hdr, 2);
  if ((flags & (FLAG_C | FLAG_K)) && s.pos != 0) run_f(s);
  return 0;
}
int meta_ad(Strobe& s, const uint8_t* d, size_t n, bool more) { int rc = begin_op(s, FLAG_M | FLAG_A, more); if (!rc) absorb(s, d, n); return rc; }
int ad(Strobe& s, const uint8_t* d, size_t n, bool more) { int rc = begin_op(s, FLAG_A, more); if (!rc) absorb(s, d, n); return rc; }
int prf(Strobe& s, uint8_t* out, size_t n, bool more) { int rc = begin_op(s, FLAG_I | FLAG_A | FLAG_C, more); if (!rc) squeeze(s, out, n); return rc; }
int key(Strobe& s, const uint8_t* d, size_t n, bool more) { int rc = begin_op(s, FLAG_A | FLAG_C, more); if (!rc) overwrite(s, d, n); return rc; }

inline void le32(uint8_t out[4], size_t v) { for (int i = 0; i < 4; ++i) out[i] = (uint8_t)(v >> (8 * i)); }

const uint64_t FR[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
bool fr_canonical_nonzero(const uint8_t b[32]) {
  uint64_t w[4]; uint64_t any = 0;
  for (int i = 0; i < 4; ++i) { uint64_t v = 0; for (int j = 7; j >= 0; --j) v = (v << 8) | b[8 * i + j]; w[i] = v; any |= v; }
  if (!any) return false;
  for (int i = 3; i >= 0; --i) { if (w[i] != FR[i]) return w[i] < FR[i]; }
  return false;
}

}  // namespace

extern "C" {

// eight Keccak-f[1600] permutations; lanes[25][8]: lane w of state k at lanes[8*w + k]
void cg1_keccak_f1600_x8(uint64_t* lanes) {
  static const bool have512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl");
  if (have512) keccak_x8_avx512(lanes); else keccak_x8_generic(lanes);
}
// Permute `live` (1..8) 200-byte sponges in place, given by address.  AVX-512: loaded, transposed and stored in
// registers; otherwise through a staging buffer and cg1_keccak_f1600_x8's baseline build.
void cg1_keccak_f1600_x8_states(uint8_t* const* states, int live) {
  static const bool have512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl");
  if (live < 1) return;
  if (live > 8) live = 8;
  if (have512) {
    uint8_t* st[8];
    for (int j = 0; j < 8; ++j) st[j] = states[j < live ? j : 0];
    keccak_x8_aos_avx512(st, live);
    return;
  }
  alignas(64) uint64_t lanes[25 * 8];
  for (int j = 0; j < 8; ++j) {
    const uint8_t* src = states[j < live ? j : 0];
    for (int w = 0; w < 25; ++w) memcpy(&lanes[8 * w + j], src + 8 * w, 8);
  }
  keccak_x8_generic(lanes);
  for (int j = 0; j < live; ++j)
    for (int w = 0; w < 25; ++w) memcpy(states[j] + 8 * w, &lanes[8 * w + j], 8);
}
// one permutation of a 200-byte state (exported for tests / microbenchmarks)
void cg1_keccak_f1600(uint8_t* state200) { keccak_f1600(state200); }

// Strobe128.new(protocol_label)   strobe.py:23-36
void cg1_strobe_new(uint8_t* state, const uint8_t* label, size_t len) {
  Strobe& s = *reinterpret_cast<Strobe*>(state);
  memset(&s, 0, sizeof s);
  const uint8_t hdr[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
  memcpy(s.st, hdr, 6);
  memcpy(s.st + 6, "STROBEv1.0.2", 12);
  keccak_f1600(s.st);
  meta_ad(s, label, len, false);
}
int cg1_strobe_meta_ad(uint8_t* state, const uint8_t* d, size_t n, int more) { return meta_ad(*reinterpret_cast<Strobe*>(state), d, n, more != 0); }
int cg1_strobe_ad(uint8_t* state, const uint8_t* d, size_t n, int more) { return ad(*reinterpret_cast<Strobe*>(state), d, n, more != 0); }
int cg1_strobe_prf(uint8_t* state, uint8_t* out, size_t n, int more) { return prf(*reinterpret_cast<Strobe*>(state), out, n, more != 0); }
int cg1_strobe_key(uint8_t* state, const uint8_t* d, size_t n, int more) { return key(*reinterpret_cast<Strobe*>(state), d, n, more != 0); }

// MerlinTranscript(label)   merlin_transcript.py:6-9
void cg1_merlin_init(uint8_t* state, const uint8_t* label, size_t len) {
  cg1_strobe_new(state, reinterpret_cast<const uint8_t*>("Merlin v1.0"), 11);
  cg1_merlin_append(state, reinterpret_cast<const uint8_t*>("dom-sep"), 7, label, len);
}
// append_message(label, message)   merlin_transcript.py:11-15
void cg1_merlin_append(uint8_t* state, const uint8_t* label, size_t llen, const uint8_t* msg, size_t mlen) {
  Strobe& s = *reinterpret_cast<Strobe*>(state);
  uint8_t dl[4]; le32(dl, mlen);
  meta_ad(s, label, llen, false);
  meta_ad(s, dl, 4, true);
  ad(s, msg, mlen, false);
}
// append_list(label, items) of equal-size items   curdleproofs_transcript.py:11-13
void cg1_merlin_append_list(uint8_t* state, const uint8_t* label, size_t llen, const uint8_t* items, size_t item_len, size_t count) {
  for (size_t i = 0; i < count; ++i) cg1_merlin_append(state, label, llen, items + i * item_len, item_len);
}
// challenge_bytes(label, length)   merlin_transcript.py:20-24
void cg1_merlin_challenge(uint8_t* state, const uint8_t* label, size_t llen, uint8_t* out, size_t n) {
  Strobe& s = *reinterpret_cast<Strobe*>(state);
  uint8_t dl[4]; le32(dl, n);
  meta_ad(s, label, llen, false);
  meta_ad(s, dl, 4, true);
  prf(s, out, n, false);
}
// get_and_append_challenge(label): rejection-sample 32 PRF bytes < r and non-zero, re-append the accepted bytes
// curdleproofs_transcript.py:15-25
void cg1_merlin_challenge_scalar(uint8_t* state, const uint8_t* label, size_t llen, uint8_t out32[32]) {
  for (;;) {
    cg1_merlin_challenge(state, label, llen, out32, 32);
    if (fr_canonical_nonzero(out32)) { cg1_merlin_append(state, label, llen, out32, 32); return; }
  }
}

}  // extern "C"
