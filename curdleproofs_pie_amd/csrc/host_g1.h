// Host-side BLS12-381 Fp / G1 (6 x 64-bit limbs, Montgomery radix 2^384, Jacobian coordinates).
//
// This is the product's *host logic*, not a fallback for the GPU path: it backs the single-element
// G1Point operators (one kernel launch per `P + Q` would be absurd), the 48-byte wire codec of single
// points, input marshalling (batch projective->affine), and the O(255)-doubling Horner tail that
// finishes a GPU MSM.  The MSM entry points are HIP-only at every size.  One data-parallel entry point may take the host's worker pool
// instead of a launch: cg1_batch_mul_add with a few hundred outputs (a fold / map of the prover), where the launch is 255 dependent
// doublings (~2.2 ms) whatever it computes -- the same records come back, byte for byte (cg1_batch_mul_add_pool, shuffle_verify.cpp;
// "batch_mul_host_max" = 0 keeps every such call on the GPU).
#pragma once
#include <cstdint>
#include <cstddef>

namespace cg1h {

struct fe { uint64_t l[6]; };          // Montgomery form, canonical (< p)
struct jac { fe X, Y, Z; };            // Z == 0  <=> identity

// ---- field
fe fe_zero();
fe fe_one();
bool fe_is_zero(const fe& a);
bool fe_is_canonical(const fe& a);             // value < p
bool fe_eq(const fe& a, const fe& b);
fe fe_add(const fe& a, const fe& b);
fe fe_sub(const fe& a, const fe& b);
fe fe_neg(const fe& a);
fe fe_mul(const fe& a, const fe& b);
fe fe_sqr(const fe& a);
fe fe_inv(const fe& a);                        // 0 -> 0
bool fe_sqrt(const fe& a, fe& out);            // false if a is a non-residue
fe fe_from_std(const uint64_t w[6]);           // standard integer (< p) -> Montgomery
void fe_to_std(const fe& a, uint64_t w[6]);    // Montgomery -> canonical standard integer
bool fe_from_le48(const uint8_t* b, fe& out);  // false if value >= p
void fe_to_le48(const fe& a, uint8_t* b);
bool fe_from_be48(const uint8_t* b, fe& out);
void fe_to_be48(const fe& a, uint8_t* b);
bool fe_lex_largest(const fe& a);              // value > (p-1)/2

// ---- group
jac jac_identity();
jac jac_generator();
bool jac_is_identity(const jac& a);
jac jac_dbl(const jac& a);
jac jac_add(const jac& a, const jac& b);
jac jac_madd(const jac& a, const fe& x2, const fe& y2);           // a + (x2, y2), affine operand not the identity
jac jac_neg(const jac& a);
bool jac_eq(const jac& a, const jac& b);
jac jac_mul(const jac& a, const uint8_t scalar_le32[32]);        // any 256-bit scalar (not reduced)
bool jac_on_curve(const jac& a);
bool jac_in_subgroup(const jac& a);
void jac_to_affine(const jac& a, fe& x, fe& y, bool& inf);
jac jac_from_affine(const fe& x, const fe& y);
jac jac_from_xyzz(const fe& X, const fe& Y, const fe& ZZ, const fe& ZZZ);
void jac_batch_to_affine(const jac* pts, size_t n, fe* xs, fe* ys, uint8_t* inf);   // one inversion

// ---- 48-byte compressed encoding (ZCash format)
void g1_compress(const jac& a, uint8_t out[48]);
void g1_compress_affine(const fe& x, const fe& y, bool inf, uint8_t out[48]);
// 0 ok; 1 bad flags; 2 x >= p; 3 not on curve; 4 not in subgroup (only when check_subgroup)
int g1_decompress(const uint8_t in[48], bool check_subgroup, jac& out);

}  // namespace cg1h
