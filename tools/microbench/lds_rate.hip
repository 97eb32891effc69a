// What the LDS of one CU delivers to ds_read_b128 / ds_read_b64_tr_b16 streams, alone and between MFMAs, by waves per SIMD
// and by the number of reads a wave keeps in flight (both MFMA phases of a SparseImageCode round
// run at ~175 B/clk of LDS reads; MI355X_MICROARCH.md gives 256 B/clk for conflict-free ds_read_b128).
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_rate tools/microbench/lds_rate.hip ; one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int kIter = 4000;

// 16 reads of 1 KB each (immediate offsets 0 .. 15 KB from the wave's base), destinations 16 distinct register quads
#define RD(i) "ds_read_b128 %" #i ", %16 offset:" #i "*1024\n\t"
#define RD16 RD(0) RD(1) RD(2) RD(3) RD(4) RD(5) RD(6) RD(7) RD(8) RD(9) RD(10) RD(11) RD(12) RD(13) RD(14) RD(15)
#define OUT16 "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7]), "=&v"(r[8]), \
              "=&v"(r[9]), "=&v"(r[10]), "=&v"(r[11]), "=&v"(r[12]), "=&v"(r[13]), "=&v"(r[14]), "=&v"(r[15])

// MODE 0: 16 reads, wait for all, repeat.  MODE 1: 16 reads with a counted wait after each one that keeps DEPTH in flight.
// MODE 2: per MFMA two reads (A, B operands), DEPTH pairs in flight, hand-placed: reads, counted wait, MFMA on operands read
//         DEPTH iterations ago.  MODE 3: the same with the G1 mix (two ds_read_b64_tr_b16 + one ds_read_b128 per MFMA).
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void bench(unsigned long long* out, float* sink) {
  extern __shared__ f32x4 lds[];   // 128 KB
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  const unsigned sh0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)lds;
  const unsigned base = sh0 + 16u * lane + 16384u * (w & 7);
  const unsigned base_tr = sh0 + 8u * lane + 16384u * (w & 7);   // 8-byte reads: a 32-lane group covers 256 contiguous bytes
  (void)base_tr;
  f32x4 r[16];
  f32x16 acc = {};
  unsigned long long t0 = 0, t1 = 0;
  if constexpr (MODE == 0) {
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < kIter; ++it) asm volatile(RD16 "s_waitcnt lgkmcnt(0)" : OUT16 : "v"(base));
    t1 = __builtin_readcyclecounter();
  } else if constexpr (MODE == 1) {
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < kIter; ++it) {
#define RDW(i) "ds_read_b128 %" #i ", %16 offset:" #i "*1024\n\ts_waitcnt lgkmcnt(%17)\n\t"
      asm volatile(RDW(0) RDW(1) RDW(2) RDW(3) RDW(4) RDW(5) RDW(6) RDW(7) RDW(8) RDW(9) RDW(10) RDW(11) RDW(12) RDW(13) RDW(14) RDW(15) ""
                   : OUT16 : "v"(base), "i"(DEPTH - 1));
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    t1 = __builtin_readcyclecounter();
  } else {
    // software pipeline, all in one asm block per 16 MFMAs: slot i of r[] holds operand pair read DEPTH/… ago
    // r[0..7] = A operands ring, r[8..15] = B operands ring (8 deep ring, DEPTH <= 7 pairs in flight)
    f32x4 a = {1, 2, 3, 4}, b = {4, 3, 2, 1};
    for (int k = 0; k < 16; ++k) r[k] = a;
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < kIter; ++it) {
      if constexpr (MODE == 2) {
        // step s (0..7): read pair into ring slot s, wait until only DEPTH pairs (2*DEPTH reads) are outstanding, MFMA on slot (s - DEPTH) & 7
#define STEP2(s, u) "ds_read_b128 %" #s ", %17 offset:" #s "*1024\n\tds_read_b128 %" #u ", %17 offset:8192+" #s "*1024\n\t" \
                    "s_waitcnt lgkmcnt(%18)\n\t"
#define MF(ai, bi) "v_mfma_f32_32x32x16_bf16 %16, %" #ai ", %" #bi ", %16\n\t"
        if constexpr (DEPTH == 2)
          asm volatile(STEP2(0, 8) MF(6, 14) STEP2(1, 9) MF(7, 15) STEP2(2, 10) MF(0, 8) STEP2(3, 11) MF(1, 9) STEP2(4, 12) MF(2, 10)
                       STEP2(5, 13) MF(3, 11) STEP2(6, 14) MF(4, 12) STEP2(7, 15) MF(5, 13) ""
                       : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                         "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+v"(acc)
                       : "v"(base), "i"(2 * DEPTH));
        else if constexpr (DEPTH == 4)
          asm volatile(STEP2(0, 8) MF(4, 12) STEP2(1, 9) MF(5, 13) STEP2(2, 10) MF(6, 14) STEP2(3, 11) MF(7, 15) STEP2(4, 12) MF(0, 8)
                       STEP2(5, 13) MF(1, 9) STEP2(6, 14) MF(2, 10) STEP2(7, 15) MF(3, 11) ""
                       : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                         "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+v"(acc)
                       : "v"(base), "i"(2 * DEPTH));
        else
          asm volatile(STEP2(0, 8) MF(2, 10) STEP2(1, 9) MF(3, 11) STEP2(2, 10) MF(4, 12) STEP2(3, 11) MF(5, 13) STEP2(4, 12) MF(6, 14)
                       STEP2(5, 13) MF(7, 15) STEP2(6, 14) MF(0, 8) STEP2(7, 15) MF(1, 9) ""
                       : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]),
                         "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), "+v"(acc)
                       : "v"(base), "i"(2 * DEPTH));
      } else {
        // the G1 mix: per MFMA two ds_read_b64_tr_b16 (the halves of the A fragment) + one ds_read_b128 (B); 3 reads per step.
        // The halves of an A quad are written separately: literal registers (v[64:95] the A ring, v[96:127] the B ring)
#define A3(s) "v[64+4*" #s ":65+4*" #s "]"
#define A3h(s) "v[66+4*" #s ":67+4*" #s "]"
#define A3q(s) "v[64+4*" #s ":67+4*" #s "]"
#define B3q(s) "v[96+4*" #s ":99+4*" #s "]"
#define STEP3(s) "ds_read_b64_tr_b16 " A3(s) ", %3 offset:" #s "*1024\n\tds_read_b64_tr_b16 " A3h(s) ", %3 offset:" #s "*1024+512\n\t" \
                 "ds_read_b128 " B3q(s) ", %1 offset:8192+" #s "*1024\n\ts_waitcnt lgkmcnt(%2)\n\t"
#define MF3(s) "v_mfma_f32_32x32x16_bf16 %0, " A3q(s) ", " B3q(s) ", %0\n\t"
#define CLOB3 "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83", \
              "v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103", \
              "v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120", \
              "v121","v122","v123","v124","v125","v126","v127"
        if constexpr (DEPTH == 1)
          asm volatile(STEP3(0) MF3(7) STEP3(1) MF3(0) STEP3(2) MF3(1) STEP3(3) MF3(2) STEP3(4) MF3(3) STEP3(5) MF3(4) STEP3(6) MF3(5)
                       STEP3(7) MF3(6) "" : "+v"(acc) : "v"(base), "i"(3 * DEPTH), "v"(base_tr) : CLOB3);
        else if constexpr (DEPTH == 2)
          asm volatile(STEP3(0) MF3(6) STEP3(1) MF3(7) STEP3(2) MF3(0) STEP3(3) MF3(1) STEP3(4) MF3(2) STEP3(5) MF3(3) STEP3(6) MF3(4)
                       STEP3(7) MF3(5) "" : "+v"(acc) : "v"(base), "i"(3 * DEPTH), "v"(base_tr) : CLOB3);
        else if constexpr (DEPTH == 3)
          asm volatile(STEP3(0) MF3(5) STEP3(1) MF3(6) STEP3(2) MF3(7) STEP3(3) MF3(0) STEP3(4) MF3(1) STEP3(5) MF3(2) STEP3(6) MF3(3)
                       STEP3(7) MF3(4) "" : "+v"(acc) : "v"(base), "i"(3 * DEPTH), "v"(base_tr) : CLOB3);
        else
          asm volatile(STEP3(0) MF3(4) STEP3(1) MF3(5) STEP3(2) MF3(6) STEP3(3) MF3(7) STEP3(4) MF3(0) STEP3(5) MF3(1) STEP3(6) MF3(2)
                       STEP3(7) MF3(3) "" : "+v"(acc) : "v"(base), "i"(3 * DEPTH), "v"(base_tr) : CLOB3);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    t1 = __builtin_readcyclecounter();
  }
  float s = 0;
  for (int k = 0; k < 16; ++k) s += r[k][0];
  for (int q = 0; q < 16; ++q) s += acc[q];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE, int DEPTH>
void run(const char* what, int threads, int ncu, unsigned long long* d_out, float* d_sink, double kb_per_iter, int mfma_per_iter) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute((const void*)bench<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  bench<MODE, DEPTH><<<ncu, threads, 128 * 1024>>>(d_out, d_sink);
  CHECK(hipEventRecord(e0));
  bench<MODE, DEPTH><<<ncu, threads, 128 * 1024>>>(d_out, d_sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long cyc = 0;
  CHECK(hipMemcpy(&cyc, d_out, 8, hipMemcpyDeviceToHost));
  const double per = (double)cyc / kIter;
  const int waves = threads / 64;
  printf("%-64s depth %2d, %d waves/SIMD: %7.1f cycles/iteration -> %6.1f B/clk/CU", what, DEPTH, waves / 4, per,
         waves * kb_per_iter * 1024.0 / per);
  if (mfma_per_iter) printf(", %5.1f cycles per MFMA and wave; %5.2f ns per MFMA and SIMD by the launch time", per / mfma_per_iter,
                            ms * 1e6 / ((double)kIter * mfma_per_iter * (waves / 4)));
  printf("  (launch %.3f ms = %.0f GB/s of LDS reads per CU)\n", ms, waves * kb_per_iter * 1024.0 * kIter / (ms * 1e6));
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  printf("%s, %d CUs\n", p.name, ncu);
  unsigned long long* d_out; float* d_sink;
  CHECK(hipMalloc(&d_out, 64)); CHECK(hipMalloc(&d_sink, sizeof(float) * 512 * ncu));
  for (int threads : {256, 512}) {
    run<0, 16>("16 x ds_read_b128, wait for all", threads, ncu, d_out, d_sink, 16, 0);
    run<1, 4>("ds_read_b128 stream, counted waits", threads, ncu, d_out, d_sink, 16, 0);
    run<1, 8>("ds_read_b128 stream, counted waits", threads, ncu, d_out, d_sink, 16, 0);
    run<1, 12>("ds_read_b128 stream, counted waits", threads, ncu, d_out, d_sink, 16, 0);
    run<1, 15>("ds_read_b128 stream, counted waits", threads, ncu, d_out, d_sink, 16, 0);
    run<2, 2>("MFMA + 2 x ds_read_b128, hand-placed", threads, ncu, d_out, d_sink, 16, 8);
    run<2, 4>("MFMA + 2 x ds_read_b128, hand-placed", threads, ncu, d_out, d_sink, 16, 8);
    run<2, 6>("MFMA + 2 x ds_read_b128, hand-placed", threads, ncu, d_out, d_sink, 16, 8);
    run<3, 1>("MFMA + 2 x ds_read_b64_tr_b16 + ds_read_b128, hand-placed", threads, ncu, d_out, d_sink, 16, 8);
    run<3, 2>("MFMA + 2 x ds_read_b64_tr_b16 + ds_read_b128, hand-placed", threads, ncu, d_out, d_sink, 16, 8);
    run<3, 3>("MFMA + 2 x ds_read_b64_tr_b16 + ds_read_b128, hand-placed", threads, ncu, d_out, d_sink, 16, 8);
    run<3, 4>("MFMA + 2 x ds_read_b64_tr_b16 + ds_read_b128, hand-placed", threads, ncu, d_out, d_sink, 16, 8);
  }
  return 0;
}
