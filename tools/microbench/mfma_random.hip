// What the matrix pipe SUSTAINS on this box: back-to-back independent bf16 MFMAs from every SIMD (operands in registers,
// one wave per SIMD, four accumulator tiles per wave) for ~2 s per row, on trivial operands (all zero) and on random ones
// (hashed bf16 bit patterns of magnitude ~1, refreshed every outer iteration so that they are not loop constants), for
// v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16.  MI355X_MICROARCH.md ("DVFS give-back"): the chip lowers its
// clock under load and random data draws more than zeros -- the datasheet's 2.5 PFLOP/s is 2.4 GHz.  SparseImageCode's
// roofline fraction reads against these rows.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_random.hip -o tools/microbench/mfma_random.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// a bf16 pair per dword: sign and mantissa random, exponent 126 / 127 (|value| in [0.5, 2))
__device__ __forceinline__ unsigned rnd_pair(unsigned h) { return (h & 0x80ff80ffu) | 0x3f003f00u | ((h >> 3) & 0x00800080u); }

template <int SHAPE, bool RANDOM>   // SHAPE 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(256) void spin(float* out, int outer, int inner, unsigned long long* clk) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int o = 0; o < outer; ++o) {
    u32x4 ua, ub;
    for (int j = 0; j < 4; ++j) {
      ua[j] = RANDOM ? rnd_pair(mix(tid * 8u + j + 0x9e3779b9u * o)) : 0u;
      ub[j] = RANDOM ? rnd_pair(mix(tid * 8u + 4 + j + 0x85ebca6bu * o)) : 0u;
    }
    const bf16x8 av = __builtin_bit_cast(bf16x8, ua), bv = __builtin_bit_cast(bf16x8, ub);
    if constexpr (SHAPE == 0) {
      f32x16 acc[4];
      for (int r = 0; r < 4; ++r)
        for (int q = 0; q < 16; ++q) acc[r][q] = 0.f;
#pragma unroll 1
      for (int i = 0; i < inner; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[r], 0, 0, 0);
      }
      for (int r = 0; r < 4; ++r)
        for (int q = 0; q < 16; ++q) s += acc[r][q];
    } else {
      f32x4 acc[8];
      for (int r = 0; r < 8; ++r)
        for (int q = 0; q < 4; ++q) acc[r][q] = 0.f;
#pragma unroll 1
      for (int i = 0; i < inner; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int r = 0; r < 8; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[r], 0, 0, 0);
      }
      for (int r = 0; r < 8; ++r)
        for (int q = 0; q < 4; ++q) s += acc[r][q];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[tid] = s;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int SHAPE, bool RANDOM>
static void run(const char* name) {
  int cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int grid = cus;   // one 4-wave workgroup per CU: one wave per SIMD
  float* out;
  unsigned long long* clk;
  (void)hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
  (void)hipMalloc(&clk, (size_t)grid * 2 * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int inner = 64, outer = 30000;
  const double mfma_per_wave = (double)outer * inner * (SHAPE == 0 ? 32 : 64);
  const double flop = SHAPE == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((spin<SHAPE, RANDOM>), dim3(grid), dim3(256), 0, 0, out, outer, inner, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[2 * 1024];
    (void)hipMemcpy(h, clk, (size_t)grid * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mhz = 0;
    for (int b = 0; b < grid; ++b) mhz += (double)h[2 * b] / (double)h[2 * b + 1] * 100.0;
    printf("%-34s rep %d  %8.1f ms  %7.1f TFLOP/s  in-kernel clock %.0f MHz\n", name, rep, ms,
           (double)grid * 4 * mfma_per_wave * flop / ms / 1e9, mhz / grid);
  }
  (void)hipFree(out);
  (void)hipFree(clk);
}

int main() {
  run<0, false>("32x32x16 bf16, zero operands");
  run<0, true>("32x32x16 bf16, random operands");
  run<1, false>("16x16x32 bf16, zero operands");
  run<1, true>("16x16x32 bf16, random operands");
  return 0;
}
