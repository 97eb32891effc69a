// Micro-benchmarks behind DESIGN.md section 3.5: what one CU of an MI355X does with the instruction mix of the
// SparseImageCode rounds.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_lds tools/microbench/mfma_lds.hip
// Each test: one workgroup per CU, N iterations of a loop body, cycles (s_memtime) per iteration on wave 0 and the
// wall-clock time of the launch (hipEvents), which calibrates the cycle counter.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <thread>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short i16x4;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int kIter = 20000;

// MODE 0: dependent MFMA chain, no memory.  1: two independent chains in one wave.  2: chain + G2-like reads (2 x b128,
// 3 ahead).  3: chain + G1-like reads (2 x tr_b64 + b128, 3 ahead).  4: like 3 plus 5 VALU instructions with a v_rcp.
// 5: G1-like reads only, no MFMA.  6: G2-like reads only.
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

// MODE 7: G1 mix + one 1 KB LDS-DMA request per two MFMAs (the kernel's ratio).  8: G1 reads + the DMA, no MFMA.
// 9: the DMA alone.  10: G1 mix + one DMA per MFMA.
template <int MODE>
__global__ __launch_bounds__(512) void bench(unsigned long long* out, float* sink, const char* src = nullptr) {
  extern __shared__ f32x4 lds[];   // 128 KB
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = f32x4{1.f, 2.f, 3.f, 4.f};
  __syncthreads();
  using lds_f32x4 = __attribute__((address_space(3))) const f32x4;
  using lds_i16x4 = __attribute__((address_space(3))) i16x4;
  const unsigned sh0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)lds;
  const unsigned row = sh0 + 16u * lane + 16384u * w;          // conflict-free b128
  const unsigned tr0 = sh0 + 16u * lane + 16384u * w, tr1 = tr0 + 8u;
  f32x16 acc = {}, acc2 = {};
  f32x4 fa[3], fb[3];
  i16x4 lo[3], hi[3];
  float side = (float)lane;
  float sv[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  auto g2_ops = [&](int k, f32x4& a, f32x4& b) {
    a = *(lds_f32x4*)(unsigned long)(row + (unsigned)(k & 7) * 1024u);
    if (MODE != 11) b = *(lds_f32x4*)(unsigned long)(row + (unsigned)(k & 7) * 1024u + 8192u);
  };
  auto g1_ops = [&](int k, i16x4& l, i16x4& h, f32x4& b) {
    l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(unsigned long)(tr0 + (unsigned)(k & 7) * 1024u));
    h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(unsigned long)(tr1 + (unsigned)(k & 7) * 1024u));
    b = *(lds_f32x4*)(unsigned long)(row + (unsigned)(k & 7) * 1024u + 8192u);
  };
  for (int k = 0; k < 3; ++k) {
    if (MODE == 2 || MODE == 6 || MODE == 11) { fb[k] = f32x4{4, 3, 2, 1}; g2_ops(k, fa[k], fb[k]); }
    if (MODE == 3 || MODE == 4 || MODE == 5 || MODE == 7 || MODE == 8 || MODE == 10) g1_ops(k, lo[k], hi[k], fb[k]);
    if (MODE < 2 || MODE == 9 || MODE >= 12) { fa[k] = f32x4{1, 2, 3, 4}; fb[k] = f32x4{4, 3, 2, 1}; lo[k] = i16x4{1, 2, 3, 4}; hi[k] = lo[k]; }
  }
  __syncthreads();
  const char* my = (const char*)(((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long)src >> 32)) << 32) |
                                (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long)src)) + 65536 * w;
  const unsigned dma_dst = __builtin_amdgcn_readfirstlane(sh0 + 65536u + 8192u * w);
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < kIter; it += 6) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int s = u % 3;
      if (MODE == 9) {
        glds16(my + 1024 * ((it + u) & 63), 16u * lane, dma_dst + 1024u * (u & 7));
        continue;
      }
      if (MODE == 2 || MODE == 6 || MODE == 11) {
        const f32x4 a = fa[s], b = fb[s];
        __builtin_amdgcn_sched_barrier(0);
        g2_ops(u + 3, fa[s], fb[s]);
        if (MODE == 2 || MODE == 11) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
        else { acc[0] += a[0] + b[0]; }
        __builtin_amdgcn_sched_barrier(0);
      } else if (MODE >= 3 && MODE < 12) {
        const i16x4 l = lo[s], h = hi[s];
        const f32x4 b = fb[s];
        __builtin_amdgcn_sched_barrier(0);
        g1_ops(u + 3, lo[s], hi[s], fb[s]);
        const auto a8 = __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
        if (MODE != 5 && MODE != 8) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
        else { acc[0] += (float)l[0] + (float)h[0] + b[0]; }
        if ((MODE == 7 || MODE == 8) && (u & 1)) glds16(my + 1024 * ((it + u) & 63), 16u * lane, dma_dst + 1024u * (u & 7));
        if (MODE == 10) glds16(my + 1024 * ((it + u) & 63), 16u * lane, dma_dst + 1024u * (u & 7));
        if (MODE == 4) side += 0.5f * side * __builtin_amdgcn_rcpf(1.0f + side * side);
        __builtin_amdgcn_sched_barrier(0);
      } else if (MODE == 12 || MODE == 13 || MODE == 14) {
        // dependent chain + N independent vector instructions per MFMA (12: 4, 13: 8) / + one global load per MFMA (14)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[s]), __builtin_bit_cast(bf16x8, fb[s]), acc, 0, 0, 0);
        if (MODE == 14) {
          const f32x4 g = *reinterpret_cast<const f32x4*>(src + 65536 * w + 1024 * ((it + u) & 63) + 16 * lane);
          side += g[0];
        } else {
#pragma unroll
          for (int k = 0; k < (MODE == 12 ? 4 : 8); ++k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sv[k]) : "v"(side));
        }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[s]), __builtin_bit_cast(bf16x8, fb[s]), acc, 0, 0, 0);
        if (MODE == 1)
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[s]), __builtin_bit_cast(bf16x8, fa[s]), acc2, 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = side;
  for (int k = 0; k < 8; ++k) r += sv[k];
  for (int q = 0; q < 16; ++q) r += acc[q] + acc2[q];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE>
void run(const char* what, int threads, int ncu, unsigned long long* d_out, float* d_sink, const char* d_src = nullptr) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipFuncSetAttribute((const void*)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  bench<MODE><<<ncu, threads, 128 * 1024>>>(d_out, d_sink, d_src);   // warm-up
  CHECK(hipEventRecord(e0));
  bench<MODE><<<ncu, threads, 128 * 1024>>>(d_out, d_sink, d_src);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long cyc = 0;
  CHECK(hipMemcpy(&cyc, d_out, 8, hipMemcpyDeviceToHost));
  const int per = (MODE == 1) ? 2 : 1;
  printf("%-58s waves/SIMD %d: %7.1f counts/iteration (%d MFMA each), launch %.3f ms -> counter %.2f GHz, %.1f ns/iteration\n", what,
         threads / 256, (double)cyc / kIter, per, ms, cyc / (ms * 1e6), ms * 1e6 / kIter);
}

// ---- `mfma_lds energy`: where the watts of a SparseImageCode round go --------------------------------
// Every component of a round's instruction mix alone and stacked, on the WHOLE chip (one 512-thread workgroup per CU) for
// ~3 s each, with the socket power and the shader clock sampled from sysfs every 20 ms while it runs:
// joules per MFMA issued = socket watts x seconds / (CUs x 8 waves x iterations).
static std::string find_hwmon() {   // the hwmon directory of THE device this process computes on (by PCI address)
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), 0) != hipSuccess) return "";
  for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
  const std::string base = std::string("/sys/bus/pci/devices/") + bus + "/hwmon";
  DIR* d = opendir(base.c_str());
  if (!d) return "";
  std::string found;
  while (dirent* e = readdir(d)) {
    if (std::strncmp(e->d_name, "hwmon", 5)) continue;
    const std::string h = base + "/" + e->d_name;
    FILE* f = fopen((h + "/freq1_input").c_str(), "r");
    if (f) {
      fclose(f);
      found = h;
      break;
    }
  }
  closedir(d);
  return found;
}
static double read_num(const std::string& path) {
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return -1;
  double v = -1;
  if (fscanf(f, "%lf", &v) != 1) v = -1;
  fclose(f);
  return v;
}

template <int MODE>
void energy_row(const char* what, int mfma_per_iter, int ncu, unsigned long long* d_out, float* d_sink, const char* d_src,
                const std::string& hw) {
  CHECK(hipFuncSetAttribute((const void*)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  std::atomic<bool> stop{false};
  std::vector<double> watts, mhz;
  std::thread sampler([&] {
    while (!stop.load()) {
      double w = read_num(hw + "/power1_input");
      if (w < 0) w = read_num(hw + "/power1_average");
      const double f = read_num(hw + "/freq1_input");
      if (w > 0) watts.push_back(w / 1e6);
      if (f > 0) mhz.push_back(f / 1e6);
      std::this_thread::sleep_for(std::chrono::milliseconds(20));
    }
  });
  for (int i = 0; i < 50; ++i) bench<MODE><<<ncu, 512, 128 * 1024>>>(d_out, d_sink, d_src);   // clocks and power settle
  CHECK(hipDeviceSynchronize());
  watts.clear();
  mhz.clear();
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 3.0) {
    for (int i = 0; i < 20; ++i) bench<MODE><<<ncu, 512, 128 * 1024>>>(d_out, d_sink, d_src);
    CHECK(hipDeviceSynchronize());
    launches += 20;
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  stop.store(true);
  sampler.join();
  double w = 0, f = 0;
  for (double x : watts) w += x;
  for (double x : mhz) f += x;
  w /= watts.empty() ? 1 : watts.size();
  f /= mhz.empty() ? 1 : mhz.size();
  const double iters = (double)launches * kIter;             // per wave
  const double ns_iter = secs * 1e9 / iters;
  const double nj_iter_wave = w * secs * 1e9 / (iters * ncu * 8.0);   // socket nanojoules per iteration of ONE wave
  printf("%-62s %7.1f ns/iteration  %6.0f W  %5.0f MHz  %6.2f nJ per iteration and wave", what, ns_iter, w, f, nj_iter_wave);
  if (mfma_per_iter) printf("  (%.2f nJ per MFMA)", nj_iter_wave / mfma_per_iter);
  printf("\n");
}

int main(int argc, char** argv) {
  if (argc > 1 && !std::strcmp(argv[1], "energy")) {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    const std::string hw = find_hwmon();
    printf("%s, %d CUs; one 512-thread workgroup per CU, ~3 s per row; power / clock: %s\n", p.name, ncu, hw.c_str());
    unsigned long long* d_out;
    float* d_sink;
    char* d_src;
    CHECK(hipMalloc(&d_out, 64));
    CHECK(hipMalloc(&d_sink, sizeof(float) * 512 * ncu));
    CHECK(hipMalloc(&d_src, 1 << 20));
    CHECK(hipMemset(d_src, 0, 1 << 20));
    {   // the socket with nothing running, 2 s
      std::this_thread::sleep_for(std::chrono::milliseconds(500));
      double w = 0;
      int n = 0;
      for (int i = 0; i < 75; ++i) {
        double x = read_num(hw + "/power1_input");
        if (x < 0) x = read_num(hw + "/power1_average");
        if (x > 0) { w += x / 1e6; ++n; }
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
      }
      printf("%-62s %25s %6.0f W\n", "idle (no kernel)", "", n ? w / n : 0.0);
    }
    energy_row<0>("MFMA chain, operands in registers", 1, ncu, d_out, d_sink, d_src, hw);
    energy_row<1>("two independent MFMA chains per wave, operands in registers", 2, ncu, d_out, d_sink, d_src, hw);
    energy_row<11>("MFMA + ONE operand from LDS (ds_read_b128)", 1, ncu, d_out, d_sink, d_src, hw);
    energy_row<2>("MFMA + BOTH operands from LDS, row reads (the kick product)", 1, ncu, d_out, d_sink, d_src, hw);
    energy_row<3>("MFMA + BOTH operands from LDS, transposed reads (the residual product)", 1, ncu, d_out, d_sink, d_src, hw);
    energy_row<7>("  ... + one 1 KB LDS-DMA request per two MFMAs (the dictionary landing)", 1, ncu, d_out, d_sink, d_src, hw);
    energy_row<4>("  ... + the prior's vector work (5 VALU with a v_rcp per MFMA)", 1, ncu, d_out, d_sink, d_src, hw);
    energy_row<6>("the kick product's LDS reads alone (no MFMA)", 0, ncu, d_out, d_sink, d_src, hw);
    energy_row<5>("the residual product's LDS reads alone (no MFMA)", 0, ncu, d_out, d_sink, d_src, hw);
    energy_row<9>("the LDS-DMA landing alone (1 KB per iteration and wave, out of L2)", 0, ncu, d_out, d_sink, d_src, hw);
    return 0;
  }
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  printf("%s, %d CUs, clock %d MHz\n", p.name, ncu, p.clockRate / 1000);
  unsigned long long* d_out; float* d_sink;
  CHECK(hipMalloc(&d_out, 64)); CHECK(hipMalloc(&d_sink, sizeof(float) * 512 * ncu));
  for (int threads : {256, 512}) {
    run<0>("dependent MFMA chain", threads, ncu, d_out, d_sink);
    run<1>("two independent MFMA chains in one wave", threads, ncu, d_out, d_sink);
    run<2>("chain + 2 x ds_read_b128 per MFMA (G2)", threads, ncu, d_out, d_sink);
    run<3>("chain + 2 x ds_read_b64_tr_b16 + ds_read_b128 (G1)", threads, ncu, d_out, d_sink);
    run<4>("G1 mix + 5 VALU with a v_rcp", threads, ncu, d_out, d_sink);
    run<5>("G1 reads only", threads, ncu, d_out, d_sink);
    run<6>("G2 reads only", threads, ncu, d_out, d_sink);
    run<11>("chain + ONE ds_read_b128 per MFMA", threads, ncu, d_out, d_sink);
  }
  char* d_src;
  CHECK(hipMalloc(&d_src, 1 << 20)); CHECK(hipMemset(d_src, 0, 1 << 20));
  for (int threads : {256, 512}) {
    run<7>("G1 mix + one 1 KB LDS-DMA per two MFMAs", threads, ncu, d_out, d_sink, d_src);
    run<10>("G1 mix + one 1 KB LDS-DMA per MFMA", threads, ncu, d_out, d_sink, d_src);
    run<8>("G1 reads + one LDS-DMA per two steps, no MFMA", threads, ncu, d_out, d_sink, d_src);
    run<9>("LDS-DMA alone (1 KB per iteration and wave)", threads, ncu, d_out, d_sink, d_src);
    run<12>("dependent chain + 4 independent v_add per MFMA", threads, ncu, d_out, d_sink, d_src);
    run<13>("dependent chain + 8 independent v_add per MFMA", threads, ncu, d_out, d_sink, d_src);
    run<14>("dependent chain + one global_load_dwordx4 per MFMA", threads, ncu, d_out, d_sink, d_src);
  }
  // one CU only: is it the chip's power limit or the CU?
  run<3>("G1 mix, ONE workgroup on the whole chip", 512, 1, d_out, d_sink);
  run<0>("dependent chain, ONE workgroup on the whole chip", 512, 1, d_out, d_sink);
  return 0;
}
