// G independent Merlin transcripts advanced IN STEP, their Keccak-f[1600] permutations batched eight at a time
// (cg1_keccak_f1600_x8: one 512-bit vector per sponge lane).  Used by the batch shuffle-verifier front-end
// (csrc/shuffle_verify.cpp): all proofs of a batch run the same sequence of transcript operations
// (curdleproofs.py:176-180, same_perm.py:91-95, grand_prod.py:175-183, ipa.py:170-176,204-212, same_scalar.py:82-99,
// same_msm.py:164-172,194-204), so the operation is common and only the sponge positions differ -- a transcript
// needs a permutation where its own position wraps (strobe.py:63-68) or where an operation forces one
// (strobe.py:103-105), and the rejection sampling of challenges (curdleproofs_transcript.py:15-25) repeats per
// transcript.  Each operation is therefore a tiny program (header bytes, label, length, data, squeeze, check) that
// every transcript executes until it needs a permutation; the pending ones are permuted together, then resumed.
// Same bytes in, same bytes out as csrc/merlin.cpp's single-transcript functions (tests/test_merlin.py).
#pragma once
#include <cstdint>
#include <cstring>

#include "../../include/curdle_g1.h"

#ifdef CG1_FE_PROFILE
#include <atomic>
#include <chrono>
extern std::atomic<long long> g_tr_prof[4];        // ns in gather, keccak x8, scatter; number of x8 calls
#define TR_NOW() std::chrono::steady_clock::now()
#define TR_ADD(i, a, b) g_tr_prof[i] += std::chrono::duration_cast<std::chrono::nanoseconds>((b) - (a)).count()
#else
#define TR_NOW() 0
#define TR_ADD(i, a, b) (void)0
#endif

namespace cg1m {

constexpr int G = 16;                              // transcripts per group (two x8 permutation batches)
constexpr int STROBE_R = 166;                      // strobe.py:4
constexpr uint8_t FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_M = 16;

struct Group {
  alignas(64) uint8_t st[G][208];                  // the 200-byte sponge of transcript k (+ slack for 8-byte accesses)
  uint8_t pos[G], pos_begin[G];
  int count = 0;                                   // transcripts in use (<= G)

  // MerlinTranscript(label) for `n` transcripts (merlin_transcript.py:6-9): identical so far, computed once
  void init(const char* label, int n) {
    uint8_t one[CG1_MERLIN_STATE_BYTES];
    cg1_merlin_init(one, reinterpret_cast<const uint8_t*>(label), strlen(label));
    count = n;
    for (int k = 0; k < G; ++k) {
      memcpy(st[k], one, 200);
      memset(st[k] + 200, 0, 8);
      pos[k] = one[200];
      pos_begin[k] = one[201];
    }
  }

  // append_message(label, data[k]) on every transcript (merlin_transcript.py:11-15)
  void append(const char* label, const uint8_t* const* data, uint32_t len) { run(label, data, nullptr, len, false); }
  // the same message on every transcript
  void append_same(const char* label, const uint8_t* data, uint32_t len) {
    const uint8_t* p[G];
    for (int k = 0; k < G; ++k) p[k] = data;
    run(label, p, nullptr, len, false);
  }
  // get_and_append_challenge(label) on every transcript: out[k] = 32 LE bytes of a canonical non-zero Fr element
  void challenge_scalar(const char* label, uint8_t (*out)[32]) { run(label, nullptr, out, 32, true); }

 private:
  uint8_t& byte_at(int k, unsigned p) { return st[k][p]; }

  // strobe.py:55-61 up to (not including) the permutation itself
  void pad(int k) {
    byte_at(k, pos[k]) ^= pos_begin[k];
    byte_at(k, pos[k] + 1u) ^= 0x04;
    byte_at(k, STROBE_R + 1) ^= 0x80;
  }
  static bool fr_canonical_nonzero(const uint8_t b[32]) {
    static const uint64_t R[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    uint64_t w[4];
    memcpy(w, b, 32);
    if (!(w[0] | w[1] | w[2] | w[3])) return false;
    for (int i = 3; i >= 0; --i)
      if (w[i] != R[i]) return w[i] < R[i];
    return false;
  }

  // One operation = one or two byte strings per transcript, absorbed a sponge lane at a time:
  //   append:     [old pos_begin, M|A] label len32 [old pos_begin, A] data                       (strobe.py:89-101, 63-68)
  //   challenge:  [.., M|A] label len32 [.., I|A|C] -> forced permutation -> squeeze 32 -> check;  retry from the start,
  //               or, accepted:  [.., M|A] label len32 [.., A] out32
  // The first byte of each 2-byte header is the transcript's pos_begin AT THAT MOMENT, so it is patched in when the
  // cursor reaches it (a permutation in between resets it).  absorb() returns true when the transcript must be
  // permuted before it can go on (padding already applied).
  struct Str {
    uint8_t b[168];             // header (<= 40) + data (<= 64) + slack for the 8-byte loads
    uint32_t total, patch2;     // header offsets: 0 and patch2
  };
  bool absorb(int k, Str& s, uint32_t& off) {
    while (off < s.total) {
      if (off == 0 || off == s.patch2) {                       // begin_op
        s.b[off] = pos_begin[k];
        pos_begin[k] = (uint8_t)(pos[k] + 1);
      }
      const unsigned p = pos[k];
      const uint32_t limit = (off < s.patch2) ? s.patch2 : s.total;   // stop at the next header: it is patched on arrival
      unsigned take = limit - off;
      if (take > (unsigned)STROBE_R - p) take = (unsigned)STROBE_R - p;
      for (unsigned i = 0; i < take; ++i) st[k][p + i] ^= s.b[off + i];
      off += take;
      pos[k] = (uint8_t)(p + take);
      if (pos[k] == STROBE_R) {
        pad(k);
        return true;
      }
    }
    return false;
  }
  static void build(Str& s, const char* label, uint32_t llen, uint32_t len, uint8_t flags2, const uint8_t* data, uint32_t dlen) {
    s.b[1] = FLAG_M | FLAG_A;
    memcpy(s.b + 2, label, llen);
    s.b[2 + llen] = (uint8_t)len; s.b[3 + llen] = (uint8_t)(len >> 8); s.b[4 + llen] = (uint8_t)(len >> 16); s.b[5 + llen] = (uint8_t)(len >> 24);
    s.patch2 = 6 + llen;
    s.b[s.patch2 + 1] = flags2;
    if (dlen) memcpy(s.b + s.patch2 + 2, data, dlen);
    s.total = s.patch2 + 2 + dlen;
  }

  // The common case of a string, without materialising it: header part (one template per operation, the two pos_begin
  // bytes patched into a local copy) and data part XORed straight from their sources in 8-byte steps.  Only possible
  // when the whole string fits before the sponge position wraps; false = use the general path.
  bool absorb_fast(int k, const Str& hdr, const uint8_t* data, uint32_t dlen) {
    const unsigned p = pos[k], hlen = hdr.total;
    if (p + hlen + dlen >= (unsigned)STROBE_R || (dlen & 7u)) return false;
    uint64_t h[6];                                            // hlen <= 2 + 32 + 4 + 2 = 40 <= 48
    memcpy(h, hdr.b, 48);
    reinterpret_cast<uint8_t*>(h)[0] = pos_begin[k];
    reinterpret_cast<uint8_t*>(h)[hdr.patch2] = (uint8_t)(p + 1);
    pos_begin[k] = (uint8_t)(p + hdr.patch2 + 1);
    uint8_t* dst = st[k] + p;
    for (unsigned i = 0; i < hlen; i += 8) {
      uint64_t v = h[i >> 3], d;
      if (hlen - i < 8) v &= (1ull << (8u * (hlen - i))) - 1ull;
      memcpy(&d, dst + i, 8);
      d ^= v;
      memcpy(dst + i, &d, 8);
    }
    dst += hlen;
    for (unsigned i = 0; i < dlen; i += 8) {
      uint64_t v, d;
      memcpy(&v, data + i, 8);
      memcpy(&d, dst + i, 8);
      d ^= v;
      memcpy(dst + i, &d, 8);
    }
    pos[k] = (uint8_t)(p + hlen + dlen);
    return true;
  }

  void run(const char* label, const uint8_t* const* data, uint8_t (*out)[32], uint32_t len, bool challenge) {
    const uint32_t llen = (uint32_t)strlen(label);
    if (llen > 32 || len > 64) return;                           // protocol labels are <= 18 bytes, messages 32 or 48
    Str hdr1, hdr2;                                              // header templates: this operation's first / second string
    if (challenge) {
      build(hdr1, label, llen, 32, FLAG_I | FLAG_A | FLAG_C, nullptr, 0);
      build(hdr2, label, llen, 32, FLAG_A, nullptr, 0);
    } else {
      build(hdr1, label, llen, len, FLAG_A, nullptr, 0);
    }
    Str str[G];               // materialised only for a transcript whose string straddles a sponge wrap
    bool built[G];
    uint8_t pc[G];            // append: 0 string, 3 done.  challenge: 0 first string, 1 squeeze + check, 2 second string, 3 done
    uint32_t off[G];
    for (int k = 0; k < count; ++k) { pc[k] = 0; off[k] = 0; built[k] = false; }
    for (;;) {
      int pending[G], np = 0;
      for (int k = 0; k < count; ++k) {
        bool need = false;
        while (pc[k] != 3 && !need) {
          if (pc[k] == 0 || pc[k] == 2) {
            const Str& hdr = pc[k] == 2 ? hdr2 : hdr1;
            const uint8_t* dptr = pc[k] == 2 ? out[k] : (challenge ? nullptr : data[k]);
            const uint32_t dlen = pc[k] == 2 ? 32u : (challenge ? 0u : len);
            if (off[k] == 0 && !built[k] && absorb_fast(k, hdr, dptr, dlen)) {
              need = false;
            } else {
              if (!built[k]) {
                memcpy(str[k].b, hdr.b, 48);
                if (dlen) memcpy(str[k].b + hdr.total, dptr, dlen);
                str[k].total = hdr.total + dlen;
                str[k].patch2 = hdr.patch2;
                built[k] = true;
              }
              need = absorb(k, str[k], off[k]);
              if (need) break;
            }
            built[k] = false;
            off[k] = 0;
            if (!challenge || pc[k] == 2) { pc[k] = 3; break; }
            pc[k] = 1;                                            // first string done: PRF forces a permutation unless pos == 0
            if (pos[k] != 0) { pad(k); need = true; }
          } else {                                                // pc == 1: squeeze 32 bytes from a fresh block, then check
            uint8_t* o = out[k];
            if (pos[k] == 0) {
              memcpy(o, st[k], 32);
              memset(st[k], 0, 32);
              pos[k] = 32;
            } else {                                              // unreachable for 32-byte outputs; kept exact (strobe.py:77-87)
              for (int i = 0; i < 32; ++i) { uint8_t& b = byte_at(k, pos[k]); o[i] = b; b = 0; ++pos[k]; }
            }
            pc[k] = fr_canonical_nonzero(o) ? 2 : 0;              // accepted: append it; else retry the first string
          }
        }
        if (need) pending[np++] = k;
      }
      if (np == 0) return;
      permute(pending, np);
      for (int i = 0; i < np; ++i) { pos[pending[i]] = 0; pos_begin[pending[i]] = 0; }
    }
  }

  // Keccak-f on the listed transcripts, eight per call, where the sponges lie
  void permute(const int* idx, int n) {
    for (int base = 0; base < n; base += 8) {
      const int m = n - base < 8 ? n - base : 8;
      uint8_t* ptr[8];
      for (int j = 0; j < m; ++j) ptr[j] = st[idx[base + j]];
      auto t0 = TR_NOW();
      cg1_keccak_f1600_x8_states(ptr, m);
      auto t1 = TR_NOW();
      TR_ADD(1, t0, t1);
#ifdef CG1_FE_PROFILE
      g_tr_prof[3] += 1;
#endif
      (void)t0; (void)t1;
    }
  }
};

}  // namespace cg1m
