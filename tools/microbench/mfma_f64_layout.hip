// Which lanes does an f64 MFMA sum?  A = x (one double per lane), B = 1 and A = 1, B = x, for v_mfma_f64_4x4x4_4b and
// v_mfma_f64_16x16x4; x = 2^(lane & 31) in one half-wave at a time, so every output is a bit set of source lanes.
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/mfma_f64_layout.hip -o tools/microbench/mfma_f64_layout.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void probe(double* out) {   // out[variant][half][lane][4]
  const int lane = threadIdx.x;
  for (int h = 0; h < 2; ++h) {
    const double x = ((lane >> 5) == h) ? (double)(1ull << (lane & 31)) : 0.0;
    const double one = 1.0;
    double d;
    d = __builtin_amdgcn_mfma_f64_4x4x4f64(x, one, 0.0, 0, 0, 0);
    out[((0 * 2 + h) * 64 + lane) * 4 + 0] = d;
    d = __builtin_amdgcn_mfma_f64_4x4x4f64(one, x, 0.0, 0, 0, 0);
    out[((1 * 2 + h) * 64 + lane) * 4 + 0] = d;
    f64x4 z = {0.0, 0.0, 0.0, 0.0};
    f64x4 q = __builtin_amdgcn_mfma_f64_16x16x4f64(x, one, z, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[((2 * 2 + h) * 64 + lane) * 4 + r] = q[r];
    q = __builtin_amdgcn_mfma_f64_16x16x4f64(one, x, z, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[((3 * 2 + h) * 64 + lane) * 4 + r] = q[r];
  }
}

int main() {
  double* d;
  const size_t n = 4 * 2 * 64 * 4;
  hipMalloc(&d, n * sizeof(double));
  hipMemset(d, 0, n * sizeof(double));
  probe<<<1, 64>>>(d);
  static double h[4 * 2 * 64 * 4];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"4x4x4_4b A=x B=1", "4x4x4_4b A=1 B=x", "16x16x4 A=x B=1", "16x16x4 A=1 B=x"};
  for (int v = 0; v < 4; ++v) {
    printf("== %s\n", names[v]);
    for (int lane = 0; lane < 64; ++lane) {
      printf("lane %2d:", lane);
      for (int r = 0; r < (v < 2 ? 1 : 4); ++r) {
        const uint64_t lo = (uint64_t)h[((v * 2 + 0) * 64 + lane) * 4 + r], hi = (uint64_t)h[((v * 2 + 1) * 64 + lane) * 4 + r];
        const uint64_t m = lo | (hi << 32);
        printf("  r%d {", r);
        for (int s = 0; s < 64; ++s)
          if (m >> s & 1) printf("%d ", s);
        printf("}");
      }
      printf("\n");
    }
  }
  return 0;
}
