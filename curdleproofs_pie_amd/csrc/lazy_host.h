// Host half of the deferred-evaluation G1Point (curdleproofs_pie_amd/py_arkworks_bls12381.py): what the Python face needs so that the
// reference's unchanged loops (`G_L[i] + G_R[i] * gamma` ipa.py:142-146, `R * k` curdleproofs.py:310-311, 585 single
// `from_compressed_bytes_unchecked` whisk_interface.py:96-106 / util.py:35-36 ...) reach batched evaluation:
//   * validating a 48-byte encoding WITHOUT the square root (flags, x < p, Jacobi symbol of x^3 + 4), so ValueError is raised where the
//     wheel raises it and y is computed later, for all pending points at once;
//   * a batch of linear combinations  out_j = sum_t c_t * B_{i_t}  over shared bases on the worker pool (interleaved width-5 NAF per
//     combination: 255 doublings shared by all of its terms) -- the path for a handful of operator results; msm_gpu.hip sends larger
//     batches through the GPU's batched MSM;
//   * pooled decompression and the endomorphism subgroup test for the bases of deferred products.
#pragma once
#include <cstddef>
#include <cstdint>
#include "host_g1.h"

namespace cg1h {

// Jacobi symbol (a / p) of a field element (the Montgomery radix 2^384 is a square, so the representation's symbol is the value's):
// +1, -1, or 0 for a == 0.  Batches of 62 "posdivsteps" (safegcd with additions only); exact.
int fe_jacobi(const fe& a);

// 0 ok (finite point or the identity: *is_identity says which); 1 bad flags; 2 x >= p; 3 not on the curve.  No square root.
int g1_validate_compressed(const uint8_t in[48], bool* is_identity);

// P = (x, y) on the curve is in the prime-order subgroup  <=>  [z^2] P == phi(P) + P  (Scott, ePrint 2021/1130; the device's
// g1_in_subgroup, g1_xyzz.h): 126 doublings + 12 additions instead of the 255 + 50 of [r]P == O.
bool g1_in_subgroup_fast(const fe& x, const fe& y);

struct aff { fe x, y; bool inf; };

// out = sum_t scalars[t] * (neg[t] ? -pts[idx[t]] : pts[idx[t]])  for t in [0, k): scalars are 32-byte little-endian integers (any
// value below 2^256; used as plain integers, as `G1Point * Scalar` does)
jac lincomb_one(const aff* pts, const uint32_t* idx, const uint8_t* neg, const uint8_t* scalars32, size_t k);

// results[j] = combination j (terms [offsets[j], offsets[j+1])) for every j of `sel` (n_sel indices; sel == nullptr: j = 0 .. n_sel - 1),
// on the process's worker pool, one combination at a time per thread.  0 ok, 1 bad argument, 3 a base is not a canonical record.
int lincomb_pool_jac(const uint8_t* bases_affine96, size_t n_bases, const uint32_t* offsets, const uint32_t* term_base, const uint8_t* term_scalars32,
                     const uint32_t* sel, size_t n_sel, jac* results, int n_threads);

}  // namespace cg1h
