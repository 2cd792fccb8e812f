// Batched Merlin v1.0 transcripts on the device (second half): one transcript per lane, every lane
// running the SAME sequence of operations on its OWN data.  Part of the single translation unit csrc/msm_gpu.hip.
//
// Stands behind merlin_transcripts/merlin_transcripts/{keccak.py:16-66, strobe.py:16-107, merlin_transcript.py:6-24} and the
// Fiat-Shamir adaptor curdleproofs/curdleproofs/curdleproofs_transcript.py:7-28 (append, challenge_bytes, and the
// rejection-sampled Fr challenge that is re-appended), like csrc/merlin.cpp on the host -- same 208-byte state blob, so a
// transcript can move between the two.  The 200-byte sponge of a lane lives in LDS ([word][lane]: conflict-free), because
// STROBE addresses it by a run-time byte position; Keccak-f[1600] loads it into registers, permutes (24 unrolled rounds)
// and stores it back.  Lanes diverge only where the protocol makes them: a sponge fills at a lane's own position, and the
// rejection sampling of a challenge (the 256-bit draw is below r with probability 0.45) repeats per lane.
#pragma once

namespace cg1merlin {

constexpr int STROBE_R = 166;
constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32;   // strobe.py:6-11
constexpr int LANES = 64;                    // threads per block

// operation kinds of the batch interpreter
enum : uint8_t { OP_APPEND = 0, OP_CHALLENGE = 1, OP_CHALLENGE_SCALAR = 2, OP_APPEND_OUT = 3,
                 // k_merlin_batch_sync / k_shuffle_front_end only (not part of the public cg1_merlin_op):
                 OP_APPEND_POINT = 4,       // 48 bytes of the data row, an encoding with the infinity flag absorbed as the canonical 0xC0 00..00
                 OP_APPEND_CONST = 5,       // bytes of the launch's constant block
                 OP_BARRIER = 16 };         // >= 16: a compute step every lane of the wave takes together (csrc/kernels_frontend.h)
struct Op {                                  // 48 bytes, the same for every lane
  uint8_t kind, label_len;
  uint16_t pad;
  uint32_t len;                              // message / challenge length in bytes
  uint32_t data_off;                         // OP_APPEND: offset of the message in the lane's data row
  uint32_t out_off;                          // challenges: offset in the lane's output row; OP_APPEND_OUT: where the message is
  uint8_t label[32];
};
static_assert(sizeof(Op) == 48, "");

__device__ __forceinline__ uint64_t rotl64(uint64_t v, int n) { return (v << n) | (v >> (64 - n)); }

struct Sponge {                              // one lane's view of the block's LDS sponge array
  uint32_t* w;                               // &lds[lane]; word i at w[i * LANES]
  uint32_t pos, pos_begin, cur_flags;

  __device__ __forceinline__ void xor_byte(uint32_t p, uint32_t v) { w[(p >> 2) * LANES] ^= v << ((p & 3u) * 8u); }
  __device__ __forceinline__ uint32_t get_byte(uint32_t p) const { return (w[(p >> 2) * LANES] >> ((p & 3u) * 8u)) & 0xffu; }
  __device__ __forceinline__ void set_byte(uint32_t p, uint32_t v) {
    const uint32_t sh = (p & 3u) * 8u;
    uint32_t& x = w[(p >> 2) * LANES];
    x = (x & ~(0xffu << sh)) | (v << sh);
  }

  __device__ __noinline__ void keccak() {    // keccak.py:16-66 (ONE copy: ~3 K instructions; STROBE reaches it from many places)
    static constexpr uint64_t RC[24] = {
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
        0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
        0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
        0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    uint64_t a[25], e[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) a[i] = (uint64_t)w[(2 * i) * LANES] | ((uint64_t)w[(2 * i + 1) * LANES] << 32);
#define CG1_KROUND(A, E, rc)                                                                                                  \
  {                                                                                                                           \
    const uint64_t c0 = A[0] ^ A[5] ^ A[10] ^ A[15] ^ A[20], c1 = A[1] ^ A[6] ^ A[11] ^ A[16] ^ A[21],                         \
                   c2 = A[2] ^ A[7] ^ A[12] ^ A[17] ^ A[22], c3 = A[3] ^ A[8] ^ A[13] ^ A[18] ^ A[23],                         \
                   c4 = A[4] ^ A[9] ^ A[14] ^ A[19] ^ A[24];                                                                  \
    const uint64_t d0 = c4 ^ rotl64(c1, 1), d1 = c0 ^ rotl64(c2, 1), d2 = c1 ^ rotl64(c3, 1), d3 = c2 ^ rotl64(c4, 1),         \
                   d4 = c3 ^ rotl64(c0, 1);                                                                                   \
    uint64_t b0, b1, b2, b3, b4;                                                                                              \
    b0 = A[0] ^ d0; b1 = rotl64(A[6] ^ d1, 44); b2 = rotl64(A[12] ^ d2, 43); b3 = rotl64(A[18] ^ d3, 21); b4 = rotl64(A[24] ^ d4, 14); \
    E[0] = b0 ^ (~b1 & b2) ^ (rc); E[1] = b1 ^ (~b2 & b3); E[2] = b2 ^ (~b3 & b4); E[3] = b3 ^ (~b4 & b0); E[4] = b4 ^ (~b0 & b1); \
    b0 = rotl64(A[3] ^ d3, 28); b1 = rotl64(A[9] ^ d4, 20); b2 = rotl64(A[10] ^ d0, 3); b3 = rotl64(A[16] ^ d1, 45); b4 = rotl64(A[22] ^ d2, 61); \
    E[5] = b0 ^ (~b1 & b2); E[6] = b1 ^ (~b2 & b3); E[7] = b2 ^ (~b3 & b4); E[8] = b3 ^ (~b4 & b0); E[9] = b4 ^ (~b0 & b1);   \
    b0 = rotl64(A[1] ^ d1, 1); b1 = rotl64(A[7] ^ d2, 6); b2 = rotl64(A[13] ^ d3, 25); b3 = rotl64(A[19] ^ d4, 8); b4 = rotl64(A[20] ^ d0, 18); \
    E[10] = b0 ^ (~b1 & b2); E[11] = b1 ^ (~b2 & b3); E[12] = b2 ^ (~b3 & b4); E[13] = b3 ^ (~b4 & b0); E[14] = b4 ^ (~b0 & b1); \
    b0 = rotl64(A[4] ^ d4, 27); b1 = rotl64(A[5] ^ d0, 36); b2 = rotl64(A[11] ^ d1, 10); b3 = rotl64(A[17] ^ d2, 15); b4 = rotl64(A[23] ^ d3, 56); \
    E[15] = b0 ^ (~b1 & b2); E[16] = b1 ^ (~b2 & b3); E[17] = b2 ^ (~b3 & b4); E[18] = b3 ^ (~b4 & b0); E[19] = b4 ^ (~b0 & b1); \
    b0 = rotl64(A[2] ^ d2, 62); b1 = rotl64(A[8] ^ d3, 55); b2 = rotl64(A[14] ^ d4, 39); b3 = rotl64(A[15] ^ d0, 41); b4 = rotl64(A[21] ^ d1, 2); \
    E[20] = b0 ^ (~b1 & b2); E[21] = b1 ^ (~b2 & b3); E[22] = b2 ^ (~b3 & b4); E[23] = b3 ^ (~b4 & b0); E[24] = b4 ^ (~b0 & b1); \
  }
#pragma unroll
    for (int round = 0; round < 24; round += 2) {
      CG1_KROUND(a, e, RC[round]);
      CG1_KROUND(e, a, RC[round + 1]);
    }
#undef CG1_KROUND
#pragma unroll
    for (int i = 0; i < 25; ++i) { w[(2 * i) * LANES] = (uint32_t)a[i]; w[(2 * i + 1) * LANES] = (uint32_t)(a[i] >> 32); }
  }

  __device__ __noinline__ void run_f() {     // strobe.py:55-61
    xor_byte(pos, pos_begin);
    xor_byte(pos + 1, 0x04);
    xor_byte(STROBE_R + 1, 0x80);
    keccak();
    pos = 0; pos_begin = 0;
  }
  __device__ void absorb1(uint32_t v) {      // strobe.py:63-68, one byte
    xor_byte(pos, v);
    if (++pos == STROBE_R) run_f();
  }
  __device__ __noinline__ void absorb(const uint8_t* d, uint32_t n) { for (uint32_t i = 0; i < n; ++i) absorb1(d[i]); }
  __device__ void begin_op(uint8_t flags, bool more) {                  // strobe.py:89-107 (the caller keeps `more` consistent)
    if (more) return;
    const uint32_t old_begin = pos_begin;
    pos_begin = pos + 1;
    cur_flags = flags;
    absorb1(old_begin);
    absorb1(flags);
    if ((flags & (FLAG_C | FLAG_K)) && pos != 0) run_f();
  }
  __device__ void meta_ad(const uint8_t* d, uint32_t n, bool more) { begin_op(FLAG_M | FLAG_A, more); absorb(d, n); }
  __device__ void ad(const uint8_t* d, uint32_t n, bool more) { begin_op(FLAG_A, more); absorb(d, n); }
  __device__ void prf(uint8_t* out, uint32_t n) {                        // strobe.py:77-87
    begin_op(FLAG_I | FLAG_A | FLAG_C, false);
    for (uint32_t i = 0; i < n; ++i) { out[i] = (uint8_t)get_byte(pos); set_byte(pos, 0); if (++pos == STROBE_R) run_f(); }
  }
  // merlin_transcript.py:11-15 / :20-24
  __device__ void frame(const uint8_t* label, uint32_t llen, uint32_t n) {
    const uint8_t dl[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    meta_ad(label, llen, false);
    meta_ad(dl, 4, true);
  }
  __device__ void append_message(const uint8_t* label, uint32_t llen, const uint8_t* msg, uint32_t n) { frame(label, llen, n); ad(msg, n, false); }
  __device__ void challenge_bytes(const uint8_t* label, uint32_t llen, uint8_t* out, uint32_t n) { frame(label, llen, n); prf(out, n); }
};

__device__ inline bool fr_canonical_nonzero(const uint8_t* b) {            // curdleproofs_transcript.py:19-23
  uint64_t wv[4], any = 0;
  for (int i = 0; i < 4; ++i) { uint64_t v = 0; for (int j = 7; j >= 0; --j) v = (v << 8) | b[8 * i + j]; wv[i] = v; any |= v; }
  if (!any) return false;
  for (int i = 3; i >= 0; --i) { if (wv[i] != cg1::H_FR[i]) return wv[i] < cg1::H_FR[i]; }
  return false;
}

// n transcripts, all starting from `init_state` (a 208-byte host blob: MerlinTranscript(label) already applied), all running
// ops[0..nops).  data: n rows of data_stride bytes; out: n rows of out_stride bytes; states_out (optional): n x 208 bytes.
__global__ void __launch_bounds__(LANES) k_merlin_batch(const uint8_t* __restrict__ init_state, const Op* __restrict__ ops, uint32_t nops,
                                                        const uint8_t* __restrict__ data, size_t data_stride, uint8_t* __restrict__ out,
                                                        size_t out_stride, uint8_t* __restrict__ states_out, uint32_t n) {
  __shared__ uint32_t lds[52 * LANES];
  const uint32_t t = blockIdx.x * LANES + threadIdx.x;
  if (t >= n) return;
  Sponge s;
  s.w = lds + threadIdx.x;
  for (int i = 0; i < 50; ++i) s.w[i * LANES] = reinterpret_cast<const uint32_t*>(init_state)[i];
  s.pos = init_state[200]; s.pos_begin = init_state[201]; s.cur_flags = init_state[202];
  const uint8_t* row = data + (size_t)t * data_stride;
  uint8_t* orow = out + (size_t)t * out_stride;
  for (uint32_t k = 0; k < nops; ++k) {
    const Op op = ops[k];                                                 // uniform: scalar loads
    if (op.kind == OP_APPEND) {
      s.append_message(op.label, op.label_len, row + op.data_off, op.len);
    } else if (op.kind == OP_APPEND_OUT) {
      s.append_message(op.label, op.label_len, orow + op.out_off, op.len);
    } else if (op.kind == OP_CHALLENGE) {
      s.challenge_bytes(op.label, op.label_len, orow + op.out_off, op.len);
    } else {                                                              // get_and_append_challenge, curdleproofs_transcript.py:15-25
      for (;;) {
        s.challenge_bytes(op.label, op.label_len, orow + op.out_off, 32);
        if (fr_canonical_nonzero(orow + op.out_off)) { s.append_message(op.label, op.label_len, orow + op.out_off, 32); break; }
      }
    }
  }
  if (states_out) {
    uint8_t* so = states_out + (size_t)t * 208;
    for (int i = 0; i < 50; ++i) reinterpret_cast<uint32_t*>(so)[i] = s.w[i * LANES];
    so[200] = (uint8_t)s.pos; so[201] = (uint8_t)s.pos_begin; so[202] = (uint8_t)s.cur_flags;
    for (int i = 203; i < 208; ++i) so[i] = 0;
  }
}

// ------------------------------------------------------------------ permutation-synchronous interpreter (round 3)
// k_merlin_batch above lets every lane call Keccak-f where ITS transcript needs it: after the first rejected challenge draw
// (probability 0.55 per draw) the lanes of a wave need their permutations at different moments and the wave executes the
// union -- ~1650 masked passes for a shuffle-shaped program whose transcripts need ~750 each -- and it reads its messages a
// byte at a time from global memory with ONE wave per SIMD: every byte load is an exposed ~0.7 us round trip (64 ms per wave).
// Here
//   * a lane's STROBE work is a RESUMABLE byte-level state machine (op index, phase, byte index: registers): every lane advances
//     until its sponge is full (or a PRF forces the permutation, or its program ends) and stops; then all lanes that stopped
//     permute TOGETHER -- one pass of the Keccak code per permutation of the slowest lane (~800 passes) -- and resume;
//   * messages are fetched in bursts of up to 48 bytes (a whole point: three independent 16-byte loads, one exposed latency),
//     absorbed four bytes at a time; the labels live in an LDS table, the operations are 16-byte records (one load each);
//     a drawn challenge stays in LDS for its range check and its re-absorption;
//   * Keccak-f is inlined at its single call site (LDS address space instead of flat accesses) with its 64-bit rotations written
//     as v_alignbit pairs (5.2 K instead of 6.9 K instructions, none of them a 64-bit shift).
// Same LDS sponge layout, same 208-byte state blob, same results as the host transcript (tests/test_merlin_gpu.py).
struct COp {                                 // 16 bytes: the kernel's own operation record (built by cg1_merlin_batch_device)
  uint32_t kind_label;                       // kind | label index << 8 | label length << 16
  uint32_t len, data_off, out_off;
};
constexpr int MAX_LABELS = 48;

__device__ __forceinline__ void rotl64p(uint32_t lo, uint32_t hi, int n, uint32_t& olo, uint32_t& ohi) {     // compile-time n in [0, 63]
  if (n == 0) { olo = lo; ohi = hi; }
  else if (n < 32) { ohi = __builtin_amdgcn_alignbit(hi, lo, 32 - n); olo = __builtin_amdgcn_alignbit(lo, hi, 32 - n); }
  else if (n == 32) { olo = hi; ohi = lo; }
  else { ohi = __builtin_amdgcn_alignbit(lo, hi, 64 - n); olo = __builtin_amdgcn_alignbit(hi, lo, 64 - n); }
}
struct U64 { uint32_t lo, hi; };
__device__ __forceinline__ U64 x2(U64 a, U64 b) { return U64{a.lo ^ b.lo, a.hi ^ b.hi}; }
__device__ __forceinline__ U64 x3(U64 a, U64 b, U64 c) {
  return U64{(uint32_t)__builtin_amdgcn_bitop3_b32(a.lo, b.lo, c.lo, 0x96), (uint32_t)__builtin_amdgcn_bitop3_b32(a.hi, b.hi, c.hi, 0x96)};
}
__device__ __forceinline__ U64 chi(U64 a, U64 b, U64 c) { return U64{a.lo ^ (~b.lo & c.lo), a.hi ^ (~b.hi & c.hi)}; }
__device__ __forceinline__ U64 rot(U64 a, int n) { U64 r; rotl64p(a.lo, a.hi, n, r.lo, r.hi); return r; }

// keccak.py:16-66 on the sponge words of one lane: w[i * LANES], i < 50.  ABSORB: the 42 words of a rate block are XOR-ed in on the way.
template <bool ABSORB = false>
__device__ __forceinline__ void keccak_words(uint32_t* w, const uint32_t* x = nullptr) {
  static constexpr uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
      0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
      0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
      0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  U64 a[25], e[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) {
    a[i].lo = w[(2 * i) * LANES]; a[i].hi = w[(2 * i + 1) * LANES];
    if (ABSORB && i < 21) { a[i].lo ^= x[2 * i]; a[i].hi ^= x[2 * i + 1]; }
  }
// theta with three-input XORs (v_bitop3_b32 0x96): a column sum is two of them per word, and D[x] = C[x-1] ^ rot(C[x+1], 1) is never
// formed -- A ^ C[x-1] ^ rot(C[x+1], 1) is one instruction per word (178 instead of 207 instructions per round)
#define CG1_D0(a) x3(a, c4, r1)
#define CG1_D1(a) x3(a, c0, r2)
#define CG1_D2(a) x3(a, c1, r3)
#define CG1_D3(a) x3(a, c2, r4)
#define CG1_D4(a) x3(a, c3, r0)
#define CG1_KR2(A, E, rc)                                                                                                     \
  {                                                                                                                           \
    const U64 c0 = x3(x3(A[0], A[5], A[10]), A[15], A[20]), c1 = x3(x3(A[1], A[6], A[11]), A[16], A[21]),                     \
              c2 = x3(x3(A[2], A[7], A[12]), A[17], A[22]), c3 = x3(x3(A[3], A[8], A[13]), A[18], A[23]),                     \
              c4 = x3(x3(A[4], A[9], A[14]), A[19], A[24]);                                                                   \
    const U64 r0 = rot(c0, 1), r1 = rot(c1, 1), r2 = rot(c2, 1), r3 = rot(c3, 1), r4 = rot(c4, 1);                            \
    U64 b0, b1, b2, b3, b4;                                                                                                   \
    b0 = CG1_D0(A[0]); b1 = rot(CG1_D1(A[6]), 44); b2 = rot(CG1_D2(A[12]), 43); b3 = rot(CG1_D3(A[18]), 21); b4 = rot(CG1_D4(A[24]), 14); \
    E[0] = chi(b0, b1, b2); E[0].lo ^= (uint32_t)(rc); E[0].hi ^= (uint32_t)((rc) >> 32);                                     \
    E[1] = chi(b1, b2, b3); E[2] = chi(b2, b3, b4); E[3] = chi(b3, b4, b0); E[4] = chi(b4, b0, b1);                           \
    b0 = rot(CG1_D3(A[3]), 28); b1 = rot(CG1_D4(A[9]), 20); b2 = rot(CG1_D0(A[10]), 3); b3 = rot(CG1_D1(A[16]), 45); b4 = rot(CG1_D2(A[22]), 61); \
    E[5] = chi(b0, b1, b2); E[6] = chi(b1, b2, b3); E[7] = chi(b2, b3, b4); E[8] = chi(b3, b4, b0); E[9] = chi(b4, b0, b1);   \
    b0 = rot(CG1_D1(A[1]), 1); b1 = rot(CG1_D2(A[7]), 6); b2 = rot(CG1_D3(A[13]), 25); b3 = rot(CG1_D4(A[19]), 8); b4 = rot(CG1_D0(A[20]), 18); \
    E[10] = chi(b0, b1, b2); E[11] = chi(b1, b2, b3); E[12] = chi(b2, b3, b4); E[13] = chi(b3, b4, b0); E[14] = chi(b4, b0, b1); \
    b0 = rot(CG1_D4(A[4]), 27); b1 = rot(CG1_D0(A[5]), 36); b2 = rot(CG1_D1(A[11]), 10); b3 = rot(CG1_D2(A[17]), 15); b4 = rot(CG1_D3(A[23]), 56); \
    E[15] = chi(b0, b1, b2); E[16] = chi(b1, b2, b3); E[17] = chi(b2, b3, b4); E[18] = chi(b3, b4, b0); E[19] = chi(b4, b0, b1); \
    b0 = rot(CG1_D2(A[2]), 62); b1 = rot(CG1_D3(A[8]), 55); b2 = rot(CG1_D4(A[14]), 39); b3 = rot(CG1_D0(A[15]), 41); b4 = rot(CG1_D1(A[21]), 2); \
    E[20] = chi(b0, b1, b2); E[21] = chi(b1, b2, b3); E[22] = chi(b2, b3, b4); E[23] = chi(b3, b4, b0); E[24] = chi(b4, b0, b1); \
  }
#pragma unroll
  for (int round = 0; round < 24; round += 2) {
    CG1_KR2(a, e, RC[round]);
    CG1_KR2(e, a, RC[round + 1]);
  }
#undef CG1_KR2
#undef CG1_D0
#undef CG1_D1
#undef CG1_D2
#undef CG1_D3
#undef CG1_D4
#pragma unroll
  for (int i = 0; i < 25; ++i) { w[(2 * i) * LANES] = a[i].lo; w[(2 * i + 1) * LANES] = a[i].hi; }
}

struct Machine {
  uint32_t* w;                               // &lds[lane]; sponge word i at w[i * LANES]
  uint32_t* drawn;                           // &lds_drawn[lane]; word j of the challenge being drawn at drawn[j * LANES]
  const uint32_t* labels;                    // LDS label table: label L word j at labels[L * 8 + j]
  const uint8_t* consts;                     // the launch's constant block (OP_APPEND_CONST; its first 48 bytes: the canonical infinity encoding)
  uint32_t pos, pos_begin, cur_flags;
  uint32_t k, ph, i, hdr, stage;             // op index, phase within the op, byte index within the phase, the two begin_op bytes, 0/1: challenge / append half of OP_CHALLENGE_SCALAR
  uint4 rec;                                 // the operation record of op `k_loaded`
  uint32_t k_loaded;

  __device__ __forceinline__ void xor_byte(uint32_t p, uint32_t v) { w[(p >> 2) * LANES] ^= v << ((p & 3u) * 8u); }
  __device__ __forceinline__ void xor_u32(uint32_t p, uint32_t v) {          // four bytes at byte position p (p + 4 <= 200)
    const uint32_t sh = (p & 3u) * 8u;
    w[(p >> 2) * LANES] ^= v << sh;
    if (sh) w[((p >> 2) + 1u) * LANES] ^= v >> (32u - sh);
  }
  __device__ __forceinline__ uint32_t take_byte(uint32_t p) {                 // PRF: read the byte, leave zero (strobe.py:81-84)
    const uint32_t sh = (p & 3u) * 8u;
    uint32_t& x = w[(p >> 2) * LANES];
    const uint32_t b = (x >> sh) & 0xffu;
    x &= ~(0xffu << sh);
    return b;
  }
  __device__ __forceinline__ void mark_f() {                                  // strobe.py:55-59, everything of run_f but the permutation
    xor_byte(pos, pos_begin);
    xor_byte(pos + 1, 0x04);
    xor_byte(STROBE_R + 1, 0x80);
  }
  // Absorb src[i .. n) from GLOBAL memory.  true = the sponge is full: permutation pending (mark_f done); resume with the same phase.
  __device__ __forceinline__ bool absorb_global(const uint8_t* __restrict__ src, uint32_t n) {
    while (i < n) {
      const uint32_t room = (uint32_t)STROBE_R - pos, left = n - i;
      const bool al = ((reinterpret_cast<uintptr_t>(src) + i) & 3u) == 0u;
      if (al && left >= 48u && room >= 48u) {                                 // a whole point: twelve independent loads, one exposed latency
        const uint32_t* q = reinterpret_cast<const uint32_t*>(src + i);
        uint32_t v[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) v[j] = q[j];
#pragma unroll
        for (int j = 0; j < 12; ++j) xor_u32(pos + 4u * j, v[j]);
        pos += 48u; i += 48u;
      } else if (al && left >= 16u && room >= 16u) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(src + i);
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = q[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) xor_u32(pos + 4u * j, v[j]);
        pos += 16u; i += 16u;
      } else if (left >= 4u && room >= 4u) {
        const uint32_t v = (uint32_t)src[i] | ((uint32_t)src[i + 1] << 8) | ((uint32_t)src[i + 2] << 16) | ((uint32_t)src[i + 3] << 24);
        xor_u32(pos, v);
        pos += 4u; i += 4u;
      } else {
        xor_byte(pos, src[i]);
        ++pos; ++i;
      }
      if (pos == (uint32_t)STROBE_R) { mark_f(); return true; }
    }
    return false;
  }
  // Absorb bytes i .. n of a little-endian word array in LDS (word j at words[j * stride])
  __device__ __forceinline__ bool absorb_lds(const uint32_t* words, uint32_t stride, uint32_t n) {
    while (i < n) {
      if ((i & 3u) == 0u && n - i >= 4u && (uint32_t)STROBE_R - pos >= 4u) {
        xor_u32(pos, words[(i >> 2) * stride]);
        pos += 4u; i += 4u;
      } else {
        xor_byte(pos, (words[(i >> 2) * stride] >> ((i & 3u) * 8u)) & 0xffu);
        ++pos; ++i;
      }
      if (pos == (uint32_t)STROBE_R) { mark_f(); return true; }
    }
    return false;
  }
  __device__ __forceinline__ bool absorb_word(uint32_t v, uint32_t n) {       // the low n <= 4 bytes of v, from byte i on
    if (i == 0u && (uint32_t)STROBE_R - pos > n) {                            // the whole unit fits with room to spare: one or two word updates
      xor_u32(pos, n == 4u ? v : (v & ((1u << (8u * n)) - 1u)));
      pos += n; i = n;
      return false;
    }
    while (i < n) {
      xor_byte(pos, (v >> (8u * i)) & 0xffu);
      ++pos; ++i;
      if (pos == (uint32_t)STROBE_R) { mark_f(); return true; }
    }
    return false;
  }
  __device__ __forceinline__ void begin(uint32_t flags) {                      // strobe.py:89-107 up to the absorb of [old_begin, flags]
    hdr = pos_begin | (flags << 8);
    pos_begin = pos + 1u;
    cur_flags = flags;
  }

  // Run this lane's program until a permutation is due (returns true; the caller permutes, then sets pos = pos_begin = 0) or the
  // program ends (returns false with done = true).
  // ... or a barrier operation is reached (returns false with blocked = true and k at that operation).
  __device__ __forceinline__ bool advance(const COp* __restrict__ ops, uint32_t nops, const uint8_t* __restrict__ row, uint8_t* __restrict__ orow, bool& done,
                                          bool& blocked) {
    for (;;) {
      if (k >= nops) { done = true; return false; }
      if (k != k_loaded) { rec = *reinterpret_cast<const uint4*>(ops + k); k_loaded = k; }      // one 16-byte load per operation
      const uint32_t kind = rec.x & 0xffu, lab = (rec.x >> 8) & 0xffu, llen = rec.x >> 16;
      if (kind >= OP_BARRIER) { blocked = true; return false; }
      const bool as_append = kind == OP_APPEND || kind == OP_APPEND_OUT || kind == OP_APPEND_POINT || kind == OP_APPEND_CONST ||
                             (kind == OP_CHALLENGE_SCALAR && stage == 1u);
      const uint32_t len = kind == OP_CHALLENGE_SCALAR ? 32u : rec.y;
      switch (ph) {
        case 0: begin(FLAG_M | FLAG_A); i = 0; ph = 1; [[fallthrough]];       // frame: meta_ad(label), merlin_transcript.py:11-15 / :20-24
        case 1: if (absorb_word(hdr, 2)) return true; i = 0; ph = 2; [[fallthrough]];
        case 2: if (absorb_lds(labels + lab * 8u, 1u, llen)) return true; i = 0; ph = 3; [[fallthrough]];
        case 3: if (absorb_word(len, 4)) return true; i = 0; ph = 4; [[fallthrough]];      // meta_ad(len as LE32, more = True)
        case 4: begin(as_append ? (uint32_t)FLAG_A : (uint32_t)(FLAG_I | FLAG_A | FLAG_C)); i = 0; ph = 5; [[fallthrough]];
        case 5: if (absorb_word(hdr, 2)) return true; i = 0; ph = as_append ? 6u : 7u; break;
        case 6: {                                                             // ad(message)
          bool full;
          if (kind == OP_CHALLENGE_SCALAR) full = absorb_lds(drawn, LANES, 32u);       // the accepted draw, still in LDS
          else if (kind == OP_APPEND_POINT) {
            const uint8_t* pt = row + rec.z;
            full = absorb_global((pt[0] & 0xC0u) == 0xC0u ? consts : pt, 48u);        // util.py:27-32: points are hashed as re-serialised
          }
          else if (kind == OP_APPEND_CONST) full = absorb_global(consts + rec.z, len);
          else full = absorb_global(kind == OP_APPEND ? row + rec.z : orow + rec.w, len);
          if (full) return true;
          ++k; ph = 0; i = 0; stage = 0;
          break;
        }
        case 7:                                                               // PRF: the C flag forces a permutation unless the sponge was just permuted
          ph = 8; i = 0;
          if (pos != 0u) { mark_f(); return true; }
          break;
        case 8: {                                                             // squeeze `len` bytes (strobe.py:77-87)
          uint8_t* o = orow + rec.w;
          // (a PRF always starts on a freshly permuted sponge -- the C flag forced it -- so pos is 0 here and whole words can be taken)
          while (i < len) {
            if (((pos | i) & 3u) == 0u && len - i >= 4u && (uint32_t)STROBE_R - pos >= 4u && (reinterpret_cast<uintptr_t>(o) & 3u) == 0u) {
              uint32_t& x = w[(pos >> 2) * LANES];
              const uint32_t v = x;
              x = 0u;
              *reinterpret_cast<uint32_t*>(o + i) = v;
              if (kind == OP_CHALLENGE_SCALAR) drawn[(i >> 2) * LANES] = v;
              pos += 4u; i += 4u;
            } else {
              const uint32_t b = take_byte(pos);
              o[i] = (uint8_t)b;
              if (kind == OP_CHALLENGE_SCALAR) {
                uint32_t& d = drawn[(i >> 2) * LANES];
                d = (i & 3u) ? (d | (b << ((i & 3u) * 8u))) : b;
              }
              ++pos; ++i;
            }
            if (pos == (uint32_t)STROBE_R) { mark_f(); return true; }
          }
          if (kind == OP_CHALLENGE_SCALAR) {                                  // curdleproofs_transcript.py:15-25: accept a canonical non-zero draw and
            uint64_t wv[4], any = 0;                                          // append it under the same label, else draw again
            for (int j = 0; j < 4; ++j) { wv[j] = (uint64_t)drawn[(2 * j) * LANES] | ((uint64_t)drawn[(2 * j + 1) * LANES] << 32); any |= wv[j]; }
            bool ok = any != 0;
            if (ok) { ok = false; for (int j = 3; j >= 0; --j) if (wv[j] != cg1::H_FR[j]) { ok = wv[j] < cg1::H_FR[j]; break; } }
            stage = ok ? 1u : 0u;
            ph = 0; i = 0;
          } else {
            ++k; ph = 0; i = 0;
          }
          break;
        }
      }
    }
  }
};

__global__ void __launch_bounds__(LANES) k_merlin_batch_sync(const uint8_t* __restrict__ init_state, const COp* __restrict__ ops, uint32_t nops,
                                                             const uint32_t* __restrict__ label_table, uint32_t nlabels,
                                                             const uint8_t* __restrict__ data, size_t data_stride, uint8_t* __restrict__ out,
                                                             size_t out_stride, uint8_t* __restrict__ states_out, uint32_t n, uint32_t* __restrict__ passes_out,
                                                             uint32_t lanes_used) {
  // lanes_used <= LANES transcripts per wave: the lanes of a wave drift apart by whole rejected draws, and the wave pays for the UNION
  // of the code paths its lanes are on -- with few lanes per wave (the chip has 1024 SIMDs, a batch of 1024 transcripts fills 16 waves)
  // a pass is cheaper and the batch simply spreads over more SIMDs
  __shared__ uint32_t lds[52 * LANES];
  __shared__ uint32_t lds_drawn[8 * LANES];
  __shared__ uint32_t lds_labels[MAX_LABELS * 8];
  const uint32_t t = blockIdx.x * lanes_used + threadIdx.x;
  const bool live = threadIdx.x < lanes_used && t < n;
  for (uint32_t j = threadIdx.x; j < nlabels * 8u; j += LANES) lds_labels[j] = label_table[j];
  Machine m;
  m.w = lds + threadIdx.x;
  m.drawn = lds_drawn + threadIdx.x;
  m.labels = lds_labels;
  m.consts = nullptr;
  for (int i = 0; i < 50; ++i) m.w[i * LANES] = reinterpret_cast<const uint32_t*>(init_state)[i];
  m.pos = init_state[200]; m.pos_begin = init_state[201]; m.cur_flags = init_state[202];
  m.k = 0; m.ph = 0; m.i = 0; m.hdr = 0; m.stage = 0; m.k_loaded = 0xffffffffu; m.rec = make_uint4(0, 0, 0, 0);
  __syncthreads();
  const uint8_t* row = data + (size_t)(live ? t : 0) * data_stride;
  uint8_t* orow = out + (size_t)(live ? t : 0) * out_stride;
  bool done = !live;
  uint32_t passes = 0;
  unsigned long long t_adv = 0, t_kec = 0;           // shader clock spent in the two halves of a pass (reported per block)
  for (;;) {
    bool needf = false;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    bool blocked = false;
    if (!done) needf = m.advance(ops, nops, row, orow, done, blocked);
    if (blocked) done = true;                        // (no compute steps in a plain transcript program: treat as its end)
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (__ballot(needf) == 0ull) break;              // a lane only stops for a permutation or at its end: nobody waits -> everyone is done
    if (needf) { keccak_words(m.w); m.pos = 0; m.pos_begin = 0; }
    const unsigned long long c2 = __builtin_amdgcn_s_memtime();
    t_adv += c1 - c0; t_kec += c2 - c1;
    ++passes;
  }
  if (passes_out && threadIdx.x == 0) {
    passes_out[4 * blockIdx.x] = passes;
    passes_out[4 * blockIdx.x + 1] = (uint32_t)(t_adv >> 8);
    passes_out[4 * blockIdx.x + 2] = (uint32_t)(t_kec >> 8);
  }
  if (states_out && live) {
    uint8_t* so = states_out + (size_t)t * 208;
    for (int i = 0; i < 50; ++i) reinterpret_cast<uint32_t*>(so)[i] = m.w[i * LANES];
    so[200] = (uint8_t)m.pos; so[201] = (uint8_t)m.pos_begin; so[202] = (uint8_t)m.cur_flags;
    for (int i = 203; i < 208; ++i) so[i] = 0;
  }
}


// ------------------------------------------------------------------ block programs (round 3; the idea and the FE's use of it: kernels_frontend.h)
// A transcript whose operation list and message lengths are the same for every lane has a STATIC byte layout between two
// permutations (a challenge always leaves the sponge at pos = its length, pos_begin = 0, however many draws it took), so the host
// cuts it into NODES (build_block_program in msm_gpu.hip) and the device hashes whole rate blocks: rows of 48 words per node and lane.
constexpr uint32_t ROW_WORDS = 48;                 // 42 content words (bytes 0 .. 167 of the sponge), info, 5 pieces
constexpr uint32_t N_PLAIN = 0, N_SQUEEZE = 1, N_END = 2, N_SQUEEZE_RAW = 3;     // (RAW: challenge_bytes without the Fr range check; k_merlin_batch_rows only)
constexpr uint32_t MAX_PIECES = 5;
// info word:  type [0,2) | barrier [2,4): 0 none, 1 X_GPROD, 2 X_DA, 3 X_FINAL | accept delta [4,6) | reject delta [6,8) | challenge slot [8,24)
// piece word: len [0,6) (0 = none) | sponge byte offset [6,14) | source [14]: 0 = the challenge just drawn, 1 = the lane's out row | source byte offset [15,32)
struct RowDesc {                                   // per (node, word): the constant part and where the proof-dependent bytes come from
  uint32_t tword;
  uint32_t src;                                    // 0 = none; else 1 | first byte [1,3) | byte count - 1 [3,5) | byte offset in the lane's data row [5,32)
};

// Rows are laid out [wave][node][lane of the wave][48 words]: the lanes of a wave stand at nearby nodes (they drift apart by rejected
// draws only: a few dozen nodes), so what a wave reads in one pass lies within a few hundred KB instead of in 64 regions 136 KB apart.
__device__ __forceinline__ size_t row_word_index(uint32_t proof, uint32_t node, uint32_t nodes, uint32_t lanes_used) {
  const uint32_t wave = proof / lanes_used, lane = proof - wave * lanes_used;
  return (((size_t)wave * nodes + node) * lanes_used + lane) * ROW_WORDS;
}

// data: n rows of `stride` bytes (the FE: a proof's L wire points).  canon48: the rows are arrays of 48-byte point encodings and one
// flagged as infinity is hashed as the wheel re-serialises it, C0 00 .. 00 (util.py:27-32).
// grid = (ceil(nodes * 48 / 256), min(n, 65535)): x walks the words of ONE transcript's rows, y the transcripts (no 64-bit division per
// word: the flat index of round 3 spent most of its ~100 instructions on t / (nodes * 48)).
__global__ void __launch_bounds__(256) k_fill_rows(const RowDesc* __restrict__ desc, uint32_t nodes, const uint8_t* __restrict__ data, size_t stride,
                                                   uint32_t canon48, uint32_t n, uint32_t lanes_used, uint32_t* __restrict__ rows) {
  const uint32_t per = nodes * ROW_WORDS;
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k >= per) return;
  const RowDesc d = desc[k];
  const uint32_t node = k / ROW_WORDS, word = k - node * ROW_WORDS;
  const uint32_t lo = (d.src >> 1) & 3u, cnt = ((d.src >> 3) & 3u) + 1u, off = d.src >> 5, k0 = canon48 ? off % 48u : 0u;
  for (uint32_t proof = blockIdx.y; proof < n; proof += gridDim.y) {
    uint32_t v = d.tword;
    if (d.src) {
      const uint8_t* src = data + (size_t)proof * stride + off;
      const bool inf = canon48 && (src[-(int)k0] & 0xC0u) == 0xC0u;
      for (uint32_t b = 0; b < cnt; ++b) {
        const uint32_t byte = inf ? (k0 + b == 0u ? 0xC0u : 0u) : (uint32_t)src[b];
        v ^= byte << (8u * (lo + b));
      }
    }
    rows[row_word_index(proof, node, nodes, lanes_used) + word] = v;
  }
}

// One late piece: `len` <= 48 bytes from the challenge just drawn (LDS, word j at drawn[j * LANES], 9 words) or from the lane's out row
// (global) XOR-ed into the sponge at byte `dst`.  All source words are fetched at once (one exposed latency), moved to the
// destination's byte alignment with v_alignbyte, masked to the piece and XOR-ed in -- no loop-carried LDS round trips.
__device__ __forceinline__ void apply_piece(uint32_t pc, uint32_t* w, const uint32_t* drawn, const uint8_t* orow) {
  const uint32_t len = pc & 63u, dst = (pc >> 6) & 255u, from_row = (pc >> 14) & 1u, so = pc >> 15;
  const uint32_t b = dst & 3u;
  // destination word j (sponge word (dst >> 2) + j) = the four source bytes from byte address  so - b + 4 j  on
  const int32_t s0 = (int32_t)so - (int32_t)b;
  const int32_t wb = s0 >> 2;                                      // (arithmetic shift: -1 when the piece starts inside destination word 0)
  const uint32_t sh = (uint32_t)s0 & 3u;
  const int32_t wmax = (int32_t)((so + len - 1u) >> 2);            // last source word that holds a byte of the piece
  uint32_t sw[14];
  if (from_row) {
    const uint32_t* g = reinterpret_cast<const uint32_t*>(orow);
#pragma unroll
    for (int j = 0; j < 14; ++j) { int32_t k = wb + j; k = k < 0 ? 0 : (k > wmax ? wmax : k); sw[j] = g[k]; }
  } else {
#pragma unroll
    for (int j = 0; j < 14; ++j) { int32_t k = wb + j; k = k < 0 ? 0 : (k > 8 ? 8 : k); sw[j] = drawn[k * LANES]; }
  }
  const uint32_t last = b + len;                                   // piece bytes within the destination words: [b, last)
  const uint32_t jl = last >> 2, pm = (1u << (8u * (last & 3u))) - 1u;
  uint32_t* d = w + (dst >> 2) * LANES;
#pragma unroll
  for (uint32_t j = 0; j < 13u; ++j) {
    if (j == 9u && last <= 36u) break;                             // (only a 48-byte piece, or one starting late in its word, reaches words 9 .. 12)
    const uint32_t v = __builtin_amdgcn_alignbyte(sw[j + 1], sw[j], sh);
    uint32_t m = j < jl ? 0xffffffffu : (j == jl ? pm : 0u);
    if (j == 0u) m &= 0xffffffffu << (8u * b);
    if (j < 9u) d[j * LANES] ^= v & m;
    else if (m) d[j * LANES] ^= v & m;
  }
}

// Keccak-f[1600] on the lane's LDS sponge with the 42 row words XOR-ed in on the way (the absorb of a whole rate block)
__device__ __forceinline__ void keccak_absorb_row(uint32_t* w, const uint32_t (&x)[ROW_WORDS]) { keccak_words<true>(w, x); }

__device__ __forceinline__ void load_row(const uint4* __restrict__ src, uint32_t (&x)[ROW_WORDS]) {
#pragma unroll
  for (int q = 0; q < 12; ++q) { const uint4 v = src[q]; x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w; }
}


// ------------------------------------------------------------------ cg1_merlin_batch_device over a block program
// The general form of the above for any operation list of the public interface (append / challenge_bytes / rejection-sampled Fr
// challenge / append-what-you-produced): no compute steps, the final sponge written as a 208-byte state blob.  Row words 43 .. 46:
// up to FOUR late pieces; word 47: the out-row byte offset of a squeeze node, or (END) pos | pos_begin << 8 | cur_flags << 16.
__global__ void __launch_bounds__(LANES) k_merlin_batch_rows(const uint8_t* __restrict__ init_state, const uint32_t* __restrict__ rows, uint32_t nodes,
                                                             uint8_t* __restrict__ out, size_t out_stride, uint8_t* __restrict__ states_out, uint32_t n,
                                                             uint32_t lanes_used, uint32_t* __restrict__ passes_out) {
  __shared__ uint32_t lds[52 * LANES];
  __shared__ uint32_t lds_drawn[9 * LANES];
  const uint32_t t = blockIdx.x * lanes_used + threadIdx.x;
  const bool live = threadIdx.x < lanes_used && t < n;
  uint32_t* w = lds + threadIdx.x;
  uint32_t* drawn = lds_drawn + threadIdx.x;
  drawn[8 * LANES] = 0u;
  for (int i = 0; i < 50; ++i) w[i * LANES] = reinterpret_cast<const uint32_t*>(init_state)[i];
  const size_t me = live ? t : 0;
  uint8_t* orow = out + me * out_stride;
  const uint4* my_rows = reinterpret_cast<const uint4*>(rows + row_word_index((uint32_t)me, 0u, nodes, lanes_used));
  const size_t row_step = (size_t)lanes_used * (ROW_WORDS / 4);
  uint32_t cur[ROW_WORDS], nxa[ROW_WORDS], nxr[ROW_WORDS];
  load_row(my_rows, cur);
  uint32_t nd = 0, passes = 0;
  bool done = !live;
  while (__ballot(!done) != 0ull) {
    if (!done) {
      const uint32_t info = cur[42], type = info & 3u;
#pragma unroll
      for (uint32_t q = 0; q < 4u; ++q) {
        const uint32_t pc = cur[43 + q];
        if ((pc & 63u) != 0u) apply_piece(pc, w, drawn, orow);
      }
      if (type == N_END) {                                                  // what is left in the open block; no permutation follows
#pragma unroll
        for (int i = 0; i < 42; ++i) w[i * LANES] ^= cur[i];
        if (states_out) {
          uint8_t* so = states_out + (size_t)t * 208;
          for (int i = 0; i < 50; ++i) reinterpret_cast<uint32_t*>(so)[i] = w[i * LANES];
          reinterpret_cast<uint32_t*>(so)[50] = cur[47] & 0xffffffu;        // pos, pos_begin, cur_flags, 0
          reinterpret_cast<uint32_t*>(so)[51] = 0u;
        }
        done = true;
      } else {
        const uint32_t da = (info >> 4) & 3u, dr = (info >> 6) & 3u;
        load_row(my_rows + (size_t)(nd + da) * row_step, nxa);
        if (type == N_SQUEEZE) load_row(my_rows + (size_t)(nd + dr) * row_step, nxr);
        keccak_absorb_row(w, cur);
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
        const uint32_t aux = cur[47];
        nd += accept ? da : dr;
#pragma unroll
        for (uint32_t j = 0; j < ROW_WORDS; ++j) cur[j] = accept ? nxa[j] : nxr[j];
        asm volatile("" ::: "memory");
        if (type == N_SQUEEZE && accept) {
          uint32_t* o = reinterpret_cast<uint32_t*>(orow + aux);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = dv[j];
        } else if (type == N_SQUEEZE_RAW) {                                 // challenge_bytes(label, len): len <= 164 bytes from pos 0, zeroed as they are taken
          const uint32_t len = (info >> 8) & 0xffu;
          for (uint32_t j = 0; 4u * j < len; ++j) {
            const uint32_t v = w[j * LANES], rem = len - 4u * j;
            if (rem >= 4u) { w[j * LANES] = 0u; *reinterpret_cast<uint32_t*>(orow + aux + 4u * j) = v; }
            else {
              w[j * LANES] = v & (0xffffffffu << (8u * rem));
              for (uint32_t b = 0; b < rem; ++b) orow[aux + 4u * j + b] = (uint8_t)(v >> (8u * b));
            }
          }
        }
      }
    }
    ++passes;
  }
  if (passes_out && threadIdx.x == 0) passes_out[blockIdx.x] = passes;
}

}  // namespace cg1merlin
