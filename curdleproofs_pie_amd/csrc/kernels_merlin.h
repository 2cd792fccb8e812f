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
enum : uint8_t { OP_APPEND = 0, OP_CHALLENGE = 1, OP_CHALLENGE_SCALAR = 2, OP_APPEND_OUT = 3 };
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

}  // namespace cg1merlin
