// Sustained matrix-pipe rates of THIS box, for reading roofline fractions against: the fp32 (v_mfma_f32_32x32x2_f32) and
// bf16 (v_mfma_f32_32x32x16_bf16) MFMA rates with every SIMD issuing back-to-back independent MFMAs for ~50 ms -- the
// datasheet peaks (157.3 / 2500 TFLOP/s) assume 2.4 GHz, the sustained clock under matrix load is lower.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <bool BF16>
__global__ __launch_bounds__(256) void spin(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int r = 0; r < 4; ++r)
    for (int q = 0; q < 16; ++q) acc[r][q] = seed * (float)(threadIdx.x + r + q);
  const float a = seed + (float)threadIdx.x, b = seed - (float)threadIdx.x;
  bf16x8 av, bv;
  for (int j = 0; j < 8; ++j) {
    av[j] = (__bf16)(a + (float)j);
    bv[j] = (__bf16)(b - (float)j);
  }
#pragma unroll 1
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (BF16) acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[r], 0, 0, 0);
        else acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int r = 0; r < 4; ++r)
    for (int q = 0; q < 16; ++q) s += acc[r][q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <bool BF16>
static void run(const char* name, double flop_per_mfma, int blocks_per_cu) {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int grid = cus * blocks_per_cu;
  float* out;
  hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = BF16 ? 60000 : 30000;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<BF16>, dim3(grid), dim3(256), 0, 0, out, iters, 1e-30f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid * 4 * iters * 32;
    printf("%s  %d waves/SIMD  rep %d  %.1f ms  %.1f TFLOP/s\n", name, blocks_per_cu, rep, ms, mfmas * flop_per_mfma / ms / 1e9);
  }
  hipFree(out);
}

int main() {
  run<false>("fp32 32x32x2 ", 2.0 * 32 * 32 * 2, 1);
  run<false>("fp32 32x32x2 ", 2.0 * 32 * 32 * 2, 2);
  run<true>("bf16 32x32x16", 2.0 * 32 * 32 * 16, 1);
  run<true>("bf16 32x32x16", 2.0 * 32 * 32 * 16, 2);
  return 0;
}
