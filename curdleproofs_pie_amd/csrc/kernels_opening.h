// Whisk tracker-opening proofs (opening.py:21-79; IsValidWhiskOpeningProof, whisk_interface.py:147-169): the batch verifier's
// FRONT-END on the device -- what csrc/shuffle_verify.cpp cg1_opening_prepare does per proof on the host.  Part of the single
// translation unit csrc/msm_gpu.hip.
//
// The transcript of an opening proof is six 48-byte appends and one challenge (opening.py:60-71: k_G, G, k_r_G, r_G, A, B under
// "tracker_opening_proof", then "tracker_opening_proof_challenge"): four Keccak passes.  It runs through cg1_merlin_batch_device's
// block program on rows k_opening_gather lays out; the two kernels here are what surrounds it:
//   k_opening_gather    wire bytes -> the five own points per proof in MSM order [k_G, k_r_G, r_G, A, B] (what k_batch_decompress
//                       reads), and the transcript's data row [those five | G] -- any encoding with the infinity flag becomes the
//                       canonical 0xC0 00 .. 00 first, because the reference hashes points after RE-serialising them (util.py:27-32)
//   k_opening_scalars   challenge c, response s, weights rho1, rho2 -> the scalars of  rho1 (s G + c k_G - A) + rho2 (s r_G + c k_r_G - B)
//                       (opening.py:73-74 as one random combination), the proof's status, and zeros for a rejected proof
//   k_fr_sum            the generator's scalar: sum of rho1 s over the batch
#pragma once
#include "fr.h"

namespace cg1open {
using cg1fr::fr;

struct Enc48 { uint32_t w[12]; };
constexpr uint32_t ROW_BYTES = 288;                       // [k_G | k_r_G | r_G | A | B | G]

__global__ void __launch_bounds__(256) k_opening_gather(const uint32_t* __restrict__ trackers, const uint32_t* __restrict__ kcs, const uint32_t* __restrict__ proofs,
                                                        Enc48 g, uint32_t n, uint32_t* __restrict__ wire, uint32_t* __restrict__ rows) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 6u * n) return;
  const uint32_t i = t / 6u, j = t - 6u * i;
  uint32_t v[12];
  if (j == 5u) {
    for (int k = 0; k < 12; ++k) v[k] = g.w[k];
  } else {
    const uint32_t* src = j == 0u ? kcs + 12ull * i
                        : j == 1u ? trackers + 24ull * i + 12
                        : j == 2u ? trackers + 24ull * i
                        : j == 3u ? proofs + 32ull * i : proofs + 32ull * i + 12;
    for (int k = 0; k < 12; ++k) v[k] = src[k];
    if ((v[0] & 0xC0u) == 0xC0u) {                        // byte 0 of the encoding is the low byte of word 0
      v[0] = 0xC0u;
      for (int k = 1; k < 12; ++k) v[k] = 0;
    }
    uint32_t* w = wire + 12ull * (5ull * i + j);
    for (int k = 0; k < 12; ++k) w[k] = v[k];
  }
  uint32_t* r = rows + (size_t)(ROW_BYTES / 4) * i + 12u * j;
  for (int k = 0; k < 12; ++k) r[k] = v[k];
}

// The weights of a batch when the caller supplies none: rho1 | rho2 of proof i = the first 32 bytes of SHAKE256(seed || le64(i)), 128 bits
// each (a forger wins with probability 2^-128 per proof; the seed is 32 fresh bytes from the OS per batch and never leaves the verifier).
// Drawing 64 bytes per proof from the OS instead costs the host more than everything the GPU does for the proof.  The same function on
// the host: cg1_opening_weights_from_seed (shuffle_verify.cpp); tests compare both with hashlib.shake_256.
struct Seed32 { uint32_t w[8]; };
__device__ __forceinline__ void weights_from_seed(uint32_t* sponge /* &lds[lane], stride LANES */, const Seed32& seed, uint32_t i, uint8_t out64[64]) {
  constexpr int L = cg1merlin::LANES;
  for (int k = 0; k < 50; ++k) sponge[k * L] = 0;
  for (int k = 0; k < 8; ++k) sponge[k * L] = seed.w[k];
  sponge[8 * L] = i;                                       // le64(i), i < 2^32
  sponge[10 * L] = 0x1Fu;                                  // SHAKE domain bits + first pad bit at byte 40
  sponge[33 * L] = 0x80000000u;                            // last pad bit: byte 135 of the 136-byte rate
  cg1merlin::keccak_words<false>(sponge);
  uint32_t* o = reinterpret_cast<uint32_t*>(out64);
  for (int k = 0; k < 16; ++k) o[k] = 0;
  for (int k = 0; k < 4; ++k) { o[k] = sponge[k * L]; o[8 + k] = sponge[(4 + k) * L]; }
}

// status codes = csrc/shuffle_verify.cpp's (CG1_SHUFFLE_BAD_SCALAR / _BAD_WEIGHT / _BAD_POINT), passed in so that this file needs no header of it
__global__ void __launch_bounds__(64) k_opening_scalars(const uint8_t* __restrict__ challenges, const uint8_t* __restrict__ proofs, const uint8_t* __restrict__ weights,
                                                        Seed32 seed, const uint8_t* __restrict__ point_status, uint32_t n, int32_t bad_scalar, int32_t bad_weight, int32_t bad_point,
                                                        uint8_t* __restrict__ scalars, uint8_t* __restrict__ g_scalars, int32_t* __restrict__ status) {
  __shared__ uint32_t sponge[50 * cg1merlin::LANES];
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  alignas(8) uint8_t wbuf[64];
  const uint8_t* wt = wbuf;
  if (weights) wt = weights + 64ull * i;
  else weights_from_seed(sponge + threadIdx.x, seed, i, wbuf);
  uint64_t* sc = reinterpret_cast<uint64_t*>(scalars + 160ull * i);
  uint64_t* gs = reinterpret_cast<uint64_t*>(g_scalars + 32ull * i);
  fr s_, r1, r2, c;
  int32_t st = 0;
  if (!cg1fr::fr_from_le32(proofs + 128ull * i + 96, s_)) st = bad_scalar;
  else if (!cg1fr::fr_from_le32(wt, r1) || !cg1fr::fr_from_le32(wt + 32, r2)) st = bad_weight;
  else {
    const uint8_t* ps = point_status + 5ull * i;
    if (ps[0] | ps[1] | ps[2] | ps[3] | ps[4]) st = bad_point;
  }
  status[i] = st;
  if (st) {
    for (int k = 0; k < 20; ++k) sc[k] = 0;
    for (int k = 0; k < 4; ++k) gs[k] = 0;
    return;
  }
  cg1fr::fr_from_le32(challenges + 32ull * i, c);           // canonical by construction (curdleproofs_transcript.py:19-23)
  uint8_t* o = scalars + 160ull * i;
  cg1fr::fr_to_le32(cg1fr::fr_mul(r1, c), o);               // k_G
  cg1fr::fr_to_le32(cg1fr::fr_mul(r2, c), o + 32);          // k_r_G
  cg1fr::fr_to_le32(cg1fr::fr_mul(r2, s_), o + 64);         // r_G
  cg1fr::fr_to_le32(cg1fr::fr_neg(r1), o + 96);             // A
  cg1fr::fr_to_le32(cg1fr::fr_neg(r2), o + 128);            // B
  cg1fr::fr_to_le32(cg1fr::fr_mul(r1, s_), g_scalars + 32ull * i);      // G
}

// out[block] = sum of the block's slice of n canonical 32-byte scalars mod r (plain values add the same way as Montgomery ones); launched
// twice: gridDim.x partial sums, then one block over those
__global__ void __launch_bounds__(256) k_fr_sum(const uint64_t* __restrict__ v, uint32_t n, uint64_t* __restrict__ out) {
  __shared__ uint64_t sh[256][4];
  fr acc = cg1fr::fr_zero();
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += 256 * gridDim.x) acc = cg1fr::fr_add(acc, fr{{v[4ull * i], v[4ull * i + 1], v[4ull * i + 2], v[4ull * i + 3]}});
  for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] = acc.l[k];
  __syncthreads();
  for (uint32_t d = 128; d >= 1; d >>= 1) {
    if (threadIdx.x < d) {
      const fr a{{sh[threadIdx.x][0], sh[threadIdx.x][1], sh[threadIdx.x][2], sh[threadIdx.x][3]}};
      const fr b{{sh[threadIdx.x + d][0], sh[threadIdx.x + d][1], sh[threadIdx.x + d][2], sh[threadIdx.x + d][3]}};
      const fr s = cg1fr::fr_add(a, b);
      for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] = s.l[k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) for (int k = 0; k < 4; ++k) out[4ull * blockIdx.x + k] = sh[0][k];
}

}  // namespace cg1open
