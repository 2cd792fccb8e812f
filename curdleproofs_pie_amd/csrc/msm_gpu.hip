// MI355X (gfx950) Pippenger MSM over BLS12-381 G1 -- the per-device pipeline context, the launch chains of the
// two regimes and the C ABI.  The kernels live in kernels_*.h (one translation unit, included below).
//
// Replaces the hot loop of the reference's compute_MSM
//   (/root/reference/curdleproofs/curdleproofs/msm_accumulator.py:6-12:  current += base * scalar)
// with a signed-digit windowed-bucket method laid out for CDNA4.  Regime A (one large MSM), in launch order:
//
//   k_prepare_points   96 B affine (std form) -> 128 B records (x,y as 14x28-bit Montgomery limbs): one aligned
//                      cache line per point, so the bucket gather touches exactly one line     [kernels_prepare_digits.h]
//   k_digits           c-bit signed digits of this rank's windows, window-major u16
//   k_part_count/scatter + k_uscan*, k_bin_sort
//                      two-level LDS partition sort by (window, bucket); no global atomics     [kernels_sort.h]
//                      (k_hist/k_scatter: global-atomic counting sort, only for n > 2^23)
//   k_scan1/2/3, k_chunk_desc, k_len_scan, k_order
//                      bucket offsets; chunks of <= L entries (L = max(8, entries/2^18), <= 4096 chunks per bucket);
//                      chunks ordered by descending length so a wave's lanes finish together
//   k_accumulate  ***  the dominant kernel: one lane per chunk, XYZZ mixed adds over gathered points [kernels_accumulate.h]
//   k_heavy_combine, k_bucket_fold
//                      re-join buckets that were cut into several chunks (skewed scalars, window-sharded ranks)
//   k_rowcol, k_small_tree
//                      2-D bucket reduction: row sums / column sums, then 1 + hb + lb masked sums per window,
//                      exported as canonical XYZZ words                                        [kernels_reduce.h]
//   host tail          one Horner over global bit positions (255 doublings + 256 additions, host_g1.cpp): one GPU
//                      lane needs ~20-30 us per dependent EC operation, a host core ~0.5 us, so the strictly serial
//                      tail belongs on the host.
// Regime B (cg1_msm_batched*): k_group_count/scatter (LDS counting sort per (msm, window)), the same chunking +
// k_accumulate, k_seg_reduce, k_group_reduce, k_msm_horner.
//
// No MFMA: this is carry-propagating big-integer arithmetic.  The bound is the VALU integer-multiply rate
// (v_mad_u64_u32), see fp28.h; bench.py reports both that and the HBM roofline the task sheet asks for.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include "g1_xyzz.h"
#include "g1_quad.h"
#include "host_g1.h"
#include "lazy_host.h"
#include "pool.h"
#include "../../include/curdle_g1.h"

namespace cg1 {

int pick_window(size_t n);

#include "kernels_records.h"
#include "fp_row.h"
#include "glv.h"
#include "kernels_prepare_digits.h"
#include "kernels_sort.h"
#include "kernels_accumulate.h"
#include "kernels_reduce.h"
#include "kernels_small.h"
#include "kernels_batch.h"
}  // namespace cg1
#include "kernels_rows.h"
#include "kernels_merlin.h"
#include "kernels_frontend.h"
#include "kernels_opening.h"
#include "host_context.h"          // Ctx: streams, helper threads, scratch buffers
#include "host_chains.h"           // planner + launch chains: regime A, k_msm_small, regime B
#include "capi_core_msm.h"         // cg1_* : host operators, context, memory, parameters, MSM entry points
#include "capi_vec_batched.h"      // resident vectors, batched normalisation, regime B
#include "capi_lincomb.h"          // deferred G1Point evaluation: cg1_lincomb_batch, cg1_batch_subgroup
#include "capi_timing_batchmul.h"  // timings / counters, cg1_batch_mul_add*
#include "capi_codec_transcripts.h"// decompression, subgroup flags, Merlin batches, opening proofs
#include "capi_frontend.h"         // the shuffle verifier front-end on the device
#include "capi_rows_probes.h"      // scalar rows, compression, synthetic scalars, probes
