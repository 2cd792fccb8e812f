// VALU instruction-rate microbenchmark for gfx950 (MI355X).
//
// Decides the big-integer field-arithmetic design for the BLS12-381 MSM
// kernels: which multiply primitive (v_mad_u64_u32 / v_mul_lo+hi /
// 24-bit mads / f64 FMA) has the best 32x32-equivalent throughput.
//
// Build:  hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o gpurun_out/ubench_valu
// Output: one line per (op, waves/SIMD): lane-ops/s chip-wide and the
//         implied cycles per wave64 instruction per SIMD at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

constexpr int CHAINS = 8;     // independent dependency chains per lane
constexpr int UNROLL = 16;    // ops per chain per loop iteration

enum Op { MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, MAD_U32_U24, MUL_HI_U32_U24, FMA_F64,
          ADD_U32, ADD_CO_U32, ADDC_CO_U32, LSHL_ADD_U64, ADD3_U32, ALIGNBIT, FMA_F32, MAD_U64_DEP1, NOPS };
static const char* op_names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24",
  "v_mul_hi_u32_u24", "v_fma_f64", "v_add_u32", "v_add_co_u32", "v_addc_co_u32", "v_lshl_add_u64",
  "v_add3_u32", "v_alignbit_b32", "v_fma_f32", "v_mad_u64_u32(1 chain)"};

template <int OP>
__global__ void __launch_bounds__(256) k_rate(uint32_t* out, int iters, uint32_t seed,
                                              unsigned long long* cyc) {
  uint32_t a = seed * (threadIdx.x + 1) | 1u, b = seed ^ (0x9e3779b9u * (blockIdx.x + 1));
  uint64_t acc[CHAINS];
  double   dacc[CHAINS];
  float    facc[CHAINS];
  uint32_t wacc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) {
    acc[c] = (uint64_t)a * (c + 3) + b; dacc[c] = 1.0 + c; facc[c] = 1.0f + c; wacc[c] = a + c;
  }
  double da = 1.0000001, db = 0.9999999;
  float fa = 1.0000001f, fb = 0.9999999f;
  uint64_t x64 = ((uint64_t)b << 32) | a;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if constexpr (OP == MAD_U64_U32)
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
        else if constexpr (OP == MAD_U64_DEP1)
          asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b) : "vcc");
        else if constexpr (OP == MUL_LO_U32)
          asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(wacc[c]) : "v"(a));
        else if constexpr (OP == MUL_HI_U32)
          asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(wacc[c]) : "v"(a));
        else if constexpr (OP == MAD_U32_U24)
          asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(wacc[c]) : "v"(a), "v"(b));
        else if constexpr (OP == MUL_HI_U32_U24)
          asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(wacc[c]) : "v"(a));
        else if constexpr (OP == FMA_F64)
          asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(dacc[c]) : "v"(da), "v"(db));
        else if constexpr (OP == FMA_F32)
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(facc[c]) : "v"(fa), "v"(fb));
        else if constexpr (OP == ADD_U32)
          asm volatile("v_add_u32 %0, %1, %0" : "+v"(wacc[c]) : "v"(a));
        else if constexpr (OP == ADD_CO_U32)
          asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(wacc[c]) : "v"(a) : "vcc");
        else if constexpr (OP == ADDC_CO_U32)
          asm volatile("v_addc_co_u32 %0, vcc, %1, %0, vcc" : "+v"(wacc[c]) : "v"(a) : "vcc");
        else if constexpr (OP == LSHL_ADD_U64)
          asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(acc[c]) : "v"(x64));
        else if constexpr (OP == ADD3_U32)
          asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(wacc[c]) : "v"(a), "v"(b));
        else if constexpr (OP == ALIGNBIT)
          asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(wacc[c]) : "v"(a));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint32_t r = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
    r ^= (uint32_t)acc[c] ^ (uint32_t)(acc[c] >> 32) ^ wacc[c] ^ (uint32_t)dacc[c] ^ (uint32_t)facc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
int run(int waves_per_simd, int iters, uint32_t* d_out, unsigned long long* d_cyc) {
  // 256 CUs x 4 SIMDs; a 256-thread block = 4 waves = 1 wave per SIMD of one CU.
  int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters / 8, 12345u, d_cyc);  // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 12345u, d_cyc);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> cyc(blocks);
  CHECK(hipMemcpy(cyc.data(), d_cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double avg_cyc = 0; for (auto c : cyc) avg_cyc += (double)c; avg_cyc /= blocks;
  double n_inst_per_wave = (double)iters * UNROLL * CHAINS;
  double lane_ops = n_inst_per_wave * 64.0 * 4.0 * blocks;
  // s_memtime ticks at the shader clock (MI355X_MICROARCH.md cycle constants).
  double cyc_per_inst_per_simd = avg_cyc / n_inst_per_wave / waves_per_simd;
  double mhz = avg_cyc / (ms * 1e3);
  printf("%-24s waves/SIMD=%d  %.3f ms  %8.2f Glane-op/s  %.2f cyc/wave-inst/SIMD (in-kernel)  clk~%.0f MHz\n",
         op_names[OP], waves_per_simd, ms, lane_ops / ms * 1e-6, cyc_per_inst_per_simd, mhz);
  return 0;
}

template <int OP>
int run_all(uint32_t* d_out, unsigned long long* d_cyc) {
  for (int w : {1, 2, 4, 8}) if (run<OP>(w, 2000, d_out, d_cyc)) return 1;
  return 0;
}

int main() {
  uint32_t* d_out; unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(uint32_t)));
  CHECK(hipMalloc(&d_cyc, 256 * 8 * sizeof(unsigned long long)));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  if (run_all<MAD_U64_U32>(d_out, d_cyc)) return 1;
  if (run_all<MAD_U64_DEP1>(d_out, d_cyc)) return 1;
  if (run_all<MUL_LO_U32>(d_out, d_cyc)) return 1;
  if (run_all<MUL_HI_U32>(d_out, d_cyc)) return 1;
  if (run_all<MAD_U32_U24>(d_out, d_cyc)) return 1;
  if (run_all<MUL_HI_U32_U24>(d_out, d_cyc)) return 1;
  if (run_all<FMA_F64>(d_out, d_cyc)) return 1;
  if (run_all<FMA_F32>(d_out, d_cyc)) return 1;
  if (run_all<ADD_U32>(d_out, d_cyc)) return 1;
  if (run_all<ADD_CO_U32>(d_out, d_cyc)) return 1;
  if (run_all<ADDC_CO_U32>(d_out, d_cyc)) return 1;
  if (run_all<LSHL_ADD_U64>(d_out, d_cyc)) return 1;
  if (run_all<ADD3_U32>(d_out, d_cyc)) return 1;
  if (run_all<ALIGNBIT>(d_out, d_cyc)) return 1;
  return 0;
}
