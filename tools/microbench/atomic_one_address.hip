// What a returning atomicAdd on ONE address costs when every workgroup of a launch does one (the jump-process launch of the
// compacted path reserves its movers' place in the next iteration's list that way: 3 906 workgroups at C4's size), against
// the same launch with the workgroups spread over 8 / 64 counters, and with three more non-returning atomics per workgroup
// (the l / f / r tallies).  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_one_address tools/microbench/atomic_one_address.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k(int* counters, int n_counters, int stride, unsigned long long* tallies, int n_tally, int* out) {
  __shared__ int base;
  if (threadIdx.x == 0) base = atomicAdd(&counters[(blockIdx.x % n_counters) * stride], 18);
  if (threadIdx.x < n_tally) atomicAdd(&tallies[threadIdx.x], 5ull);
  __syncthreads();
  if (threadIdx.x < 18) out[(size_t)blockIdx.x * 18 + threadIdx.x] = base + threadIdx.x;
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 3906;
  int *counters, *out;
  unsigned long long* tallies;
  CHECK(hipMalloc(&counters, 64 * 64 * sizeof(int)));
  CHECK(hipMalloc(&tallies, 64));
  CHECK(hipMalloc(&out, (size_t)grid * 18 * sizeof(int)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int n_tally : {0, 3})
    for (int nc : {1, 8, 64}) {
      CHECK(hipMemset(counters, 0, 64 * 64 * sizeof(int)));
      CHECK(hipMemset(tallies, 0, 64));
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, counters, nc, 64, tallies, n_tally, out);
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, counters, nc, 64, tallies, n_tally, out);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("%d workgroups, %2d counters (256 B apart), %d tallies: %.1f us per launch\n", grid, nc, n_tally, ms * 1e3 / 50);
    }
  return 0;
}
