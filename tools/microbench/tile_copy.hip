// What a streaming pass over two (N x 256 B) row matrices costs -- read X and V, write X' and V', nothing in between -- by
// the shape of the pass: how many bytes a wavefront has in flight, waves per workgroup, grid (one tile per wave or a
// persistent grid), and whether the stores of a tile are issued while the next tile's loads are in flight.  The ceiling of
// the elementwise trajectory launches (C4's trajectory launch moves 1.03 GB per iteration).
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/tile_copy tools/microbench/tile_copy.hip ; usage: /tmp/tile_copy [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) float f32x4;

// a wave moves tiles of CHUNKS KB per matrix (CHUNKS 16-byte loads per lane); WAVES waves per workgroup.
// PERSIST: the grid is sized by the host and waves stride over the tiles, else one tile per wave.
// PIPE (persistent only): the next tile's loads are issued before this tile's stores.
template <int CHUNKS, int WAVES, bool PERSIST, bool PIPE>
__global__ __launch_bounds__(WAVES * 64) void copy2(const f32x4* __restrict__ X, const f32x4* __restrict__ V, f32x4* __restrict__ Xo,
                                                    f32x4* __restrict__ Vo, long n_tiles, int gap) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * WAVES + (threadIdx.x >> 6), n_waves = (long)gridDim.x * WAVES;
  f32x4 a[CHUNKS], b[CHUNKS];
  auto load = [&](long t) {
    const f32x4* x = X + t * (CHUNKS * 64) + lane;
    const f32x4* v = V + t * (CHUNKS * 64) + lane;
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) a[i] = x[i * 64];
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) b[i] = v[i * 64];
  };
  auto store = [&](long t, const f32x4 (&p)[CHUNKS], const f32x4 (&q)[CHUNKS]) {
    f32x4* x = Xo + t * (CHUNKS * 64) + lane;
    f32x4* v = Vo + t * (CHUNKS * 64) + lane;
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) x[i * 64] = p[i];
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) v[i * 64] = q[i];
  };
  // `gap` passes of dependent multiply-adds over the tile between its loads and its stores (8 * CHUNKS vector instructions
  // each): the work a trajectory does while the wave has nothing in flight
  auto work = [&]() {
    for (int g = 0; g < gap; ++g) {
#pragma unroll
      for (int i = 0; i < CHUNKS; ++i) {
        a[i] = a[i] * 1.0000001f + b[i];
        b[i] = b[i] * 0.9999999f + a[i];
      }
    }
  };
  if constexpr (!PERSIST) {
    if (wave < n_tiles) {
      load(wave);
      work();
      store(wave, a, b);
    }
  } else if constexpr (!PIPE) {
    for (long t = wave; t < n_tiles; t += n_waves) {
      load(t);
      work();
      store(t, a, b);
    }
  } else {
    long t = wave;
    if (t < n_tiles) load(t);
    while (t < n_tiles) {
      f32x4 p[CHUNKS], q[CHUNKS];
#pragma unroll
      for (int i = 0; i < CHUNKS; ++i) {
        p[i] = a[i];
        q[i] = b[i];
      }
      const long nt = t + n_waves;
      if (nt < n_tiles) load(nt);
      store(t, p, q);
      t = nt;
    }
  }
}

template <int CHUNKS, int WAVES, bool PERSIST, bool PIPE>
void run(const char* tag, const f32x4* X, const f32x4* V, f32x4* Xo, f32x4* Vo, long rows, int waves_per_cu, int gap = 0) {
  const long n_tiles = rows * 16 / (CHUNKS * 64);   // rows of 256 B = 16 chunks
  int cus = 0;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const long grid = PERSIST ? (long)cus * waves_per_cu / WAVES : (n_tiles + WAVES - 1) / WAVES;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy2<CHUNKS, WAVES, PERSIST, PIPE>), dim3((unsigned)grid), dim3(WAVES * 64), 0, 0, X, V, Xo, Vo, n_tiles, gap);
  CHECK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((copy2<CHUNKS, WAVES, PERSIST, PIPE>), dim3((unsigned)grid), dim3(WAVES * 64), 0, 0, X, V, Xo, Vo, n_tiles, gap);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, bytes = (double)rows * 256 * 4;
  printf("%-58s %7.1f us  %5.2f TB/s\n", tag, us, bytes / us * 1e-6);
}

int main(int argc, char** argv) {
  const long rows = argc > 1 ? atol(argv[1]) : 1000000 / 64 * 64;
  const size_t bytes = (size_t)rows * 256;
  f32x4 *X, *V, *Xo, *Vo;
  CHECK(hipMalloc(&X, bytes));
  CHECK(hipMalloc(&V, bytes));
  CHECK(hipMalloc(&Xo, bytes));
  CHECK(hipMalloc(&Vo, bytes));
  CHECK(hipMemset(X, 1, bytes));
  CHECK(hipMemset(V, 2, bytes));
  printf("rows %ld x 256 B, two matrices in, two out (%.2f GB per pass)\n", rows, bytes * 4e-9);
  run<4, 4, false, false>("4 KB per wave and matrix, 4 waves/WG, tile per wave", X, V, Xo, Vo, rows, 0);
  run<16, 1, false, false>("16 KB per wave and matrix, 1 wave/WG, tile per wave", X, V, Xo, Vo, rows, 0);
  run<16, 4, false, false>("16 KB per wave and matrix, 4 waves/WG, tile per wave", X, V, Xo, Vo, rows, 0);
  run<8, 1, false, false>("8 KB per wave and matrix, 1 wave/WG, tile per wave", X, V, Xo, Vo, rows, 0);
  run<4, 1, false, false>("4 KB per wave and matrix, 1 wave/WG, tile per wave", X, V, Xo, Vo, rows, 0);
  run<16, 1, true, false>("16 KB, persistent 8 waves/CU", X, V, Xo, Vo, rows, 8);
  run<16, 1, true, true>("16 KB, persistent 8 waves/CU, loads ahead of stores", X, V, Xo, Vo, rows, 8);
  run<8, 1, true, true>("8 KB, persistent 8 waves/CU, loads ahead of stores", X, V, Xo, Vo, rows, 8);
  run<8, 1, true, true>("8 KB, persistent 16 waves/CU, loads ahead of stores", X, V, Xo, Vo, rows, 16);
  run<4, 1, true, true>("4 KB, persistent 16 waves/CU, loads ahead of stores", X, V, Xo, Vo, rows, 16);
  run<4, 1, true, true>("4 KB, persistent 32 waves/CU, loads ahead of stores", X, V, Xo, Vo, rows, 32);
  run<4, 4, true, false>("4 KB, 4 waves/WG, persistent 12 waves/CU", X, V, Xo, Vo, rows, 12);
  run<4, 4, true, false>("4 KB, 4 waves/WG, persistent 24 waves/CU", X, V, Xo, Vo, rows, 24);
  printf("-- waves per CU a streaming pass needs\n");
  run<16, 1, true, false>("16 KB, persistent 2 waves/CU", X, V, Xo, Vo, rows, 2);
  run<16, 1, true, false>("16 KB, persistent 4 waves/CU", X, V, Xo, Vo, rows, 4);
  run<16, 1, true, false>("16 KB, persistent 6 waves/CU", X, V, Xo, Vo, rows, 6);
  run<8, 1, true, false>("8 KB, persistent 8 waves/CU", X, V, Xo, Vo, rows, 8);
  run<8, 1, true, false>("8 KB, persistent 16 waves/CU", X, V, Xo, Vo, rows, 16);
  printf("-- with work between the loads and the stores of a tile (vector instructions per tile)\n");
  for (int gap : {2, 8, 16, 32}) {
    char tag[128];
    snprintf(tag, sizeof tag, "16 KB, tile per wave, %d instr", gap * 8 * 16);
    run<16, 1, false, false>(tag, X, V, Xo, Vo, rows, 0, gap);
    snprintf(tag, sizeof tag, "16 KB, persistent 8 waves/CU, %d instr", gap * 8 * 16);
    run<16, 1, true, false>(tag, X, V, Xo, Vo, rows, 8, gap);
    snprintf(tag, sizeof tag, "8 KB, persistent 16 waves/CU, %d instr", gap * 8 * 8);
    run<8, 1, true, false>(tag, X, V, Xo, Vo, rows, 16, gap);
    snprintf(tag, sizeof tag, "4 KB, persistent 32 waves/CU, %d instr", gap * 8 * 4);
    run<4, 1, true, false>(tag, X, V, Xo, Vo, rows, 32, gap);
  }
  return 0;
}
