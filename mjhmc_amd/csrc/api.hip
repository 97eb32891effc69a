// C ABI of libmjhmc_hip.so (include/mjhmc_hip.h): handle management, host<->device re-tiling,
// launch sequencing of the fused jump kernels, integer bookkeeping, the sample ring.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "autocor.hpp"
#include "handles.hpp"

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
static thread_local std::string g_err;

int mjhmc_fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
static int fail(int code, const std::string& msg) { return mjhmc_fail(code, msg); }

// The call block (handles.hpp): [Control, kCtlBytes][flf_counts, counts_bytes(cap)][stats, cap x 4 x 8 bytes]
constexpr size_t kCtlBytes = 64;
static_assert(sizeof(Control) <= kCtlBytes, "the failure flag's slot of the call block");
static size_t counts_bytes(int cap) { return (((size_t)cap + 1) * sizeof(int) + 63) / 64 * 64; }
static int ensure_call_block(mjhmc_sampler* s, int rows) {
  if (s->call_block && s->stats_cap >= rows) return 0;
  const int cap = std::max(rows, 1024);
  if (s->call_block) {   // (calls end with a stream sync: nothing in flight reads the old block)
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipFree(s->call_block));
    s->call_block = nullptr;
    s->ctl = nullptr;
    s->flf_counts = nullptr;
    s->stats = nullptr;
    s->stats_cap = 0;
  }
  const size_t total = kCtlBytes + counts_bytes(cap) + (size_t)cap * 4 * sizeof(long long);
  HIPCHK(hipMalloc((void**)&s->call_block, total));
  HIPCHK(hipMemsetAsync(s->call_block, 0, total, s->stream));
  s->ctl = (Control*)s->call_block;
  s->flf_counts = (int*)(s->call_block + kCtlBytes);
  s->stats = (long long*)(s->call_block + kCtlBytes + counts_bytes(cap));
  s->stats_cap = cap;
  return 0;
}
// the start of a call: failure flag, counters and the first `rows` tallies zeroed by ONE fill
static int zero_call(mjhmc_sampler* s, int rows) {
  TRY(ensure_call_block(s, rows));
  HIPCHK(hipMemsetAsync(s->call_block, 0, kCtlBytes + counts_bytes(s->stats_cap) + (size_t)rows * 4 * sizeof(long long), s->stream));
  return 0;
}

static int pow2ceil(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// The A/B switches of the measurement tools and of the launch-strategy tests (MJHMC_NO_FUSE, MJHMC_NO_COMPACT,
// MJHMC_NO_SPLIT, MJHMC_SPLIT_PARTS, MJHMC_NO_FSPEC, MJHMC_NO_ROWS, MJHMC_NO_LIST_CARRY, MJHMC_FUSE_BELOW, MJHMC_NO_BLOCK_DECIDE, MJHMC_NO_WPP, MJHMC_NO_QUAD, MJHMC_CHUNKS_PER_LANE,
// MJHMC_SIC_COPIES) and the failure-placing hook MJHMC_DEBUG_POISON exist only in libmjhmc_hip_test.so (built with
// -DMJHMC_TEST_HOOKS, `make test_hooks`).  The shipped library consults no environment variable on the sampling path:
// the only ones it reads at all name libraries to dlopen (MJHMC_RCCL_LIB; hipRTC / hipFFT by their sonames).
#ifdef MJHMC_TEST_HOOKS
static const char* test_env(const char* name) { return std::getenv(name); }
#else
static const char* test_env(const char*) { return nullptr; }
#endif

static int ab_flags() {
  return (test_env("MJHMC_NO_BLOCK_DECIDE") ? kAbNoBlockDecide : 0) | (test_env("MJHMC_NO_WPP") ? kAbNoWpp : 0) |
         (test_env("MJHMC_NO_QUAD") ? kAbNoQuad : 0) | (test_env("MJHMC_NO_ROWS") ? kAbNoRows : 0) |
         (test_env("MJHMC_NO_RELAY") ? kAbNoRelay : 0) | (test_env("MJHMC_FORCE_RELAY") ? kAbForceRelay : 0);
}

static int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// lanes-per-particle / elements-per-lane selection (see elementwise.hpp header comment)
// fused MarkovJumpHMC launches of this sampler run in row form (elementwise.hpp: mjhmc_fused_rows_kernel)
static bool fused_rows(const mjhmc_sampler* s) {
  const int kind = s->en->ep.kind;
  // (the mixture's force is a division per coordinate: its row form pays through the relay kernel only -- 0.500 ms against
  // the group form's 0.535 at 32 x 10^6, L = 15, and 0.734 in the one-wave row kernel -- so it takes the row form where the
  // relay runs, launch_fused_rows: from MMGaussF::kRelayMinL leapfrog steps up)
  if (kind == MJHMC_E_MM_GAUSS && s->L < MMGaussF<double>::kRelayMinL && !test_env("MJHMC_FORCE_RELAY")) return false;
  return (kind == MJHMC_E_FUNNEL_NEAL || kind == MJHMC_E_FUNNEL_REF || kind == MJHMC_E_MM_GAUSS) && s->dtype == MJHMC_F64 &&
         s->sh.E == 8 &&
         fused_rows_shape(s->mode, s->sh.logG, ab_flags());
}

int pick_shape(int D, int dtype, Shape* out) {
  const int esize = dtype == MJHMC_F64 ? 8 : 4;
  const int VEC = 16 / esize;
  Shape s;
  s.esize = esize;
  s.pitch = (D + VEC - 1) / VEC * VEC;
  s.CH = s.pitch / VEC;
  int C, G;
  if (s.CH <= 1) {
    C = 1;
    G = 1;
  } else {
    C = 4;
    if (const char* force = test_env("MJHMC_CHUNKS_PER_LANE")) {  // perf experiments
      const int f = std::atoi(force);
      C = f == 8 ? 8 : (f == 1 ? 1 : 4);
    }
    G = pow2ceil((s.CH + C - 1) / C);
    if (G > 64) {
      C = 8;
      G = pow2ceil((s.CH + 7) / 8);
    }
    if (G > 64) {
      // more dims than 64 lanes x 16 elements hold: float64 samplers of the built-in elementwise energies take the
      // multi-pass path (host_energy.hip: state in HBM between the substeps); others have no such form
      // (float32 state: the same path on float64 storage, the state rounded to float32 wherever it is written)
      s.esize = 8;
      s.pitch = (D + 1) / 2 * 2;
      s.CH = s.pitch / 2;
      s.E = 0;
      s.logG = 0;
      s.wide = true;
      s.round32 = dtype != MJHMC_F64;
      *out = s;
      return 0;
    }
  }
  s.E = C * VEC;
  s.logG = ilog2(G);
  *out = s;
  return 0;
}

// row layout and path of a sampler (or of a one-off evaluation) of energy e with state dtype `dtype`.  On return *dtype is
// the dtype the state is STORED in (float64 where float32 was asked for rows only the multi-pass path handles: sh->round32)
static int shape_for(const mjhmc_energy* e, int* dtype, Shape* sh) {
  if (e->is_pot()) {
    if (*dtype != MJHMC_F64 && *dtype != MJHMC_F32) return fail(MJHMC_ERR_UNSUPPORTED, "PRODUCT_OF_T runs with float32 or float64 state");
    if (*dtype == MJHMC_F64 || e->pot_big()) {
      // the reference's own arithmetic: float64 HMCState arrays around the float32 force (distributions.py:408-415,
      // hmc_state.py:29-38); ndims > 512: the blocked force evaluation exists on the multi-pass path only
      *sh = Shape{0, 0, e->pot_dim, e->pot_dim / 2, 8, true};
      sh->round32 = *dtype != MJHMC_F64;
    } else {
      *sh = Shape{0, 0, e->pot_dim, e->pot_dim / 4, 4};
    }
  } else if (e->is_sic()) {
    if (*dtype != MJHMC_BF16 && *dtype != MJHMC_F32)
      return fail(MJHMC_ERR_UNSUPPORTED, "SPARSE_CODE runs with BF16 state (the benchmark's) or F32 state (the reference's)");
    // a particle row = n_patches x n_coeffs elements of the state's type (the matrix-core operands are bf16 either way)
    *sh = *dtype == MJHMC_BF16 ? Shape{0, 0, e->ep.ndims, e->ep.ndims / 8, 2} : Shape{0, 0, e->ep.ndims, e->ep.ndims / 4, 4};
  } else {
    TRY(pick_shape(e->ep.ndims, *dtype, sh));
  }
  if (sh->round32) *dtype = MJHMC_F64;
  return 0;
}

// float32-valued float64 rows (Shape::round32)
__global__ void round32_kernel(double* __restrict__ a, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = (double)(float)a[i];
}
int round_rows32(mjhmc_sampler* s, void* rows) {
  const int64_t n = s->Npad * (int64_t)s->sh.pitch;
  hipLaunchKernelGGL(round32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, (double*)rows, n);
  HIPCHK(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// small kernels: re-tiling between the reference's (ndims, nparticles) float64 host layout and
// the particle-major device layout; integer bookkeeping
// ---------------------------------------------------------------------------------------------

// mjhmc_draw_from / mjhmc_min_idx: the jump process's helpers on caller arrays (mjhmc/misc/utils.py:15-50)
__global__ void draw_from_kernel(const double* __restrict__ rates, const double* __restrict__ e, int64_t n,
                                 double* __restrict__ out, long long* first_bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool bad = false;
  out[i] = wait_time(rates[i], e[i], bad);
  if (bad) atomicMin(first_bad, (long long)i);
}
__global__ void min_idx_kernel(const double* __restrict__ draws, int k, int64_t n, int32_t* __restrict__ which) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int w = 0;
  double best = draws[j];
  for (int r = 1; r < k; ++r) {   // np.argmin: the first minimum; the first NaN wins outright
    const double d = draws[(size_t)r * n + j];
    if (!(best != best) && (d < best || d != d)) {
      w = r;
      best = d;
    }
  }
  which[j] = w;
}

// src (D, N) float64 row-major  ->  dst [N][pitch] T ; only d < D is written
template <typename T>
__global__ void to_particle_major(const double* __restrict__ src, T* __restrict__ dst, int D, int64_t N, int pitch) {
  __shared__ double tile[32][33];
  const int64_t p0 = (int64_t)blockIdx.x * 32;
  const int d0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int dd = threadIdx.y + 8 * i;
    const int d = d0 + dd;
    const int64_t p = p0 + threadIdx.x;
    if (d < D && p < N) tile[dd][threadIdx.x] = src[(size_t)d * N + p];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pp = threadIdx.y + 8 * i;
    const int64_t p = p0 + pp;
    const int d = d0 + threadIdx.x;
    if (d < D && p < N) dst[(size_t)p * pitch + d] = (T)tile[threadIdx.x][pp];
  }
}

// dst[d*rs + k*cs + off] = src[row(k)][d],  row(k) = idx ? idx[k] : k
template <typename T>
__global__ void to_dim_major(const T* __restrict__ src, const int64_t* __restrict__ idx, double* __restrict__ dst,
                             int D, int64_t ncols, int pitch, int64_t rs, int64_t cs, int64_t off) {
  __shared__ double tile[32][33];
  const int64_t k0 = (int64_t)blockIdx.x * 32;
  const int d0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kk = threadIdx.y + 8 * i;
    const int64_t k = k0 + kk;
    const int d = d0 + threadIdx.x;
    if (d < D && k < ncols) {
      const int64_t row = idx ? idx[k] : k;
      tile[kk][threadIdx.x] = (double)src[(size_t)row * pitch + d];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int dd = threadIdx.y + 8 * i;
    const int d = d0 + dd;
    const int64_t k = k0 + threadIdx.x;
    if (d < D && k < ncols) dst[(size_t)d * rs + (size_t)k * cs + off] = tile[threadIdx.x][dd];
  }
}

template <typename T>
__global__ void widen_vec(const T* __restrict__ src, double* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (double)src[i];
}

// cache_active[i] = !isnan(H_flf[i])
template <typename T>
__global__ void flags_from_hflf(const T* __restrict__ h, uint8_t* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (h[i] == h[i]) ? 1 : 0;
}

// block-reduced shifted moments of the valid elements of `rows` particle rows
template <typename T>
__global__ void moments_kernel(const T* __restrict__ src, int64_t nrows_pad_per_slot, int64_t nrows_per_slot, int nslots,
                               int D, int pitch, double shift, double* out) {
  double s1 = 0.0, s2 = 0.0;
  const int64_t per_slot = nrows_per_slot * D;
  const int64_t total = per_slot * nslots;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t slot = i / per_slot, rem = i - slot * per_slot;
    const int64_t row = rem / D;
    const int d = (int)(rem - row * D);
    const double v = (double)src[(size_t)(slot * nrows_pad_per_slot + row) * pitch + d] - shift;
    s1 += v;
    s2 += v * v;
  }
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  __shared__ double sm[2][16];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sm[0][w] = s1;
    sm[1][w] = s2;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    double t = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sm[threadIdx.x][k];
    atomicAdd(&out[threadIdx.x], t);
  }
}

template <typename T>
__global__ void narrow_vec(const double* __restrict__ src, T* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (T)src[i];
}

// indices of the particles that satisfy a predicate (inverse-L cache cold; moved by R), in arbitrary order.  Each block scans
// kColdChunk particles, collects its hits in LDS and reserves its stretch of the list with ONE global atomic
// (a global atomic per wave serialised on the single counter: 178 us for 10^6 particles, this form ~10 us).
constexpr int kColdChunk = 4096;
template <typename T>
struct ColdCache {  // H_flf is NaN; the scan also writes the copy of H_flf that the inverse-L pass completes
  const T* h;
  T* copy;
  __device__ bool operator()(int64_t p) const {
    const T v = h[p];
    copy[p] = v;
    return v != v;
  }
};
template <class Pred>
__global__ __launch_bounds__(1024) void compact_list_kernel(const Pred pred, int64_t N, const Control* ctl,
                                                            int* __restrict__ list, int* __restrict__ count) {
  __shared__ int hits[kColdChunk];
  __shared__ int n_hits, base;
  if (ctl->failed) return;
  if (threadIdx.x == 0) n_hits = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < kColdChunk / 1024; ++k) {
    const int64_t p = (int64_t)blockIdx.x * kColdChunk + k * 1024 + threadIdx.x;
    const bool cold = p < N && pred(p);
    const unsigned long long mask = __ballot(cold);
    if (mask) {
      int at = 0;
      if (lane == 0) at = atomicAdd(&n_hits, __popcll(mask));
      at = __shfl(at, 0);
      if (cold) hits[at + __popcll(mask & ((1ull << lane) - 1ull))] = (int)p;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) base = atomicAdd(count, n_hits);
  __syncthreads();
  for (int i = threadIdx.x; i < n_hits; i += 1024) list[base + i] = hits[i];
}

// ---------------------------------------------------------------------------------------------
// dispatch over energies / dtypes
// ---------------------------------------------------------------------------------------------
template <typename T>
static int dispatch_jump(int kind, const JumpArgs<T>& a, const EnergyParams& ep, int E, hipStream_t st);

template <>
int dispatch_jump<double>(int kind, const JumpArgs<double>& a, const EnergyParams& ep, int E, hipStream_t st) {
  switch (kind) {
    case MJHMC_E_ISO_GAUSS: iso_jump_f64(a, ep, E, st); break;
    case MJHMC_E_DIAG_GAUSS: diag_jump_f64(a, ep, E, st); break;
    case MJHMC_E_ROUGH_WELL: rough_jump_f64(a, ep, E, st); break;
    case MJHMC_E_MM_GAUSS: mm_jump_f64(a, ep, E, st); break;
    case MJHMC_E_FUNNEL_NEAL: funnel_neal_jump_f64(a, ep, E, st); break;
    case MJHMC_E_FUNNEL_REF: funnel_ref_jump_f64(a, ep, E, st); break;
    default: return fail(MJHMC_ERR_UNSUPPORTED, "energy kind has no fused jump kernel yet");
  }
  return 0;
}
template <>
int dispatch_jump<float>(int kind, const JumpArgs<float>& a, const EnergyParams& ep, int E, hipStream_t st) {
  switch (kind) {
    case MJHMC_E_ISO_GAUSS: iso_jump_f32(a, ep, E, st); break;
    case MJHMC_E_DIAG_GAUSS: diag_jump_f32(a, ep, E, st); break;
    case MJHMC_E_ROUGH_WELL: rough_jump_f32(a, ep, E, st); break;
    case MJHMC_E_MM_GAUSS: mm_jump_f32(a, ep, E, st); break;
    case MJHMC_E_FUNNEL_NEAL: funnel_neal_jump_f32(a, ep, E, st); break;
    case MJHMC_E_FUNNEL_REF: funnel_ref_jump_f32(a, ep, E, st); break;
    default: return fail(MJHMC_ERR_UNSUPPORTED, "energy kind has no fused jump kernel yet");
  }
  return 0;
}

template <typename T>
static int dispatch_leap(int kind, const LeapArgs<T>& a, const EnergyParams& ep, int E, hipStream_t st);
template <>
int dispatch_leap<double>(int kind, const LeapArgs<double>& a, const EnergyParams& ep, int E, hipStream_t st) {
  switch (kind) {
    case MJHMC_E_ISO_GAUSS: iso_leap_f64(a, ep, E, st); break;
    case MJHMC_E_DIAG_GAUSS: diag_leap_f64(a, ep, E, st); break;
    case MJHMC_E_ROUGH_WELL: rough_leap_f64(a, ep, E, st); break;
    case MJHMC_E_MM_GAUSS: mm_leap_f64(a, ep, E, st); break;
    case MJHMC_E_FUNNEL_NEAL: funnel_neal_leap_f64(a, ep, E, st); break;
    case MJHMC_E_FUNNEL_REF: funnel_ref_leap_f64(a, ep, E, st); break;
    default: return fail(MJHMC_ERR_UNSUPPORTED, "the stand-alone leapfrog operator exists for the elementwise energies");
  }
  return 0;
}
template <>
int dispatch_leap<float>(int kind, const LeapArgs<float>& a, const EnergyParams& ep, int E, hipStream_t st) {
  switch (kind) {
    case MJHMC_E_ISO_GAUSS: iso_leap_f32(a, ep, E, st); break;
    case MJHMC_E_DIAG_GAUSS: diag_leap_f32(a, ep, E, st); break;
    case MJHMC_E_ROUGH_WELL: rough_leap_f32(a, ep, E, st); break;
    case MJHMC_E_MM_GAUSS: mm_leap_f32(a, ep, E, st); break;
    case MJHMC_E_FUNNEL_NEAL: funnel_neal_leap_f32(a, ep, E, st); break;
    case MJHMC_E_FUNNEL_REF: funnel_ref_leap_f32(a, ep, E, st); break;
    default: return fail(MJHMC_ERR_UNSUPPORTED, "the stand-alone leapfrog operator exists for the elementwise energies");
  }
  return 0;
}

template <typename T>
static int dispatch_step(int kind, const TrajArgs<T>* ta, const JumpDecideArgs<T>* da, const EnergyParams& ep, int E, hipStream_t st);
template <>
int dispatch_step<double>(int kind, const TrajArgs<double>* ta, const JumpDecideArgs<double>* da, const EnergyParams& ep, int E,
                          hipStream_t st) {
  switch (kind) {
    case MJHMC_E_ISO_GAUSS: iso_step_f64(ta, da, ep, E, st); break;
    case MJHMC_E_DIAG_GAUSS: diag_step_f64(ta, da, ep, E, st); break;
    case MJHMC_E_ROUGH_WELL: rough_step_f64(ta, da, ep, E, st); break;
    case MJHMC_E_MM_GAUSS: mm_step_f64(ta, da, ep, E, st); break;
    case MJHMC_E_FUNNEL_NEAL: funnel_neal_step_f64(ta, da, ep, E, st); break;
    case MJHMC_E_FUNNEL_REF: funnel_ref_step_f64(ta, da, ep, E, st); break;
    default: return fail(MJHMC_ERR_UNSUPPORTED, "energy kind has no trajectory / jump-process kernel");
  }
  return 0;
}
template <>
int dispatch_step<float>(int kind, const TrajArgs<float>* ta, const JumpDecideArgs<float>* da, const EnergyParams& ep, int E,
                         hipStream_t st) {
  switch (kind) {
    case MJHMC_E_ISO_GAUSS: iso_step_f32(ta, da, ep, E, st); break;
    case MJHMC_E_DIAG_GAUSS: diag_step_f32(ta, da, ep, E, st); break;
    case MJHMC_E_ROUGH_WELL: rough_step_f32(ta, da, ep, E, st); break;
    case MJHMC_E_MM_GAUSS: mm_step_f32(ta, da, ep, E, st); break;
    case MJHMC_E_FUNNEL_NEAL: funnel_neal_step_f32(ta, da, ep, E, st); break;
    case MJHMC_E_FUNNEL_REF: funnel_ref_step_f32(ta, da, ep, E, st); break;
    default: return fail(MJHMC_ERR_UNSUPPORTED, "energy kind has no trajectory / jump-process kernel");
  }
  return 0;
}

template <typename T>
static int dispatch_eval(int kind, const EvalArgs<T>& a, const EnergyParams& ep, int E, hipStream_t st);
template <>
int dispatch_eval<double>(int kind, const EvalArgs<double>& a, const EnergyParams& ep, int E, hipStream_t st) {
  switch (kind) {
    case MJHMC_E_ISO_GAUSS: iso_eval_f64(a, ep, E, st); break;
    case MJHMC_E_DIAG_GAUSS: diag_eval_f64(a, ep, E, st); break;
    case MJHMC_E_ROUGH_WELL: rough_eval_f64(a, ep, E, st); break;
    case MJHMC_E_MM_GAUSS: mm_eval_f64(a, ep, E, st); break;
    case MJHMC_E_FUNNEL_NEAL: funnel_neal_eval_f64(a, ep, E, st); break;
    case MJHMC_E_FUNNEL_REF: funnel_ref_eval_f64(a, ep, E, st); break;
    default: return fail(MJHMC_ERR_UNSUPPORTED, "energy kind has no evaluation kernel yet");
  }
  return 0;
}
template <>
int dispatch_eval<float>(int kind, const EvalArgs<float>& a, const EnergyParams& ep, int E, hipStream_t st) {
  switch (kind) {
    case MJHMC_E_ISO_GAUSS: iso_eval_f32(a, ep, E, st); break;
    case MJHMC_E_DIAG_GAUSS: diag_eval_f32(a, ep, E, st); break;
    case MJHMC_E_ROUGH_WELL: rough_eval_f32(a, ep, E, st); break;
    case MJHMC_E_MM_GAUSS: mm_eval_f32(a, ep, E, st); break;
    case MJHMC_E_FUNNEL_NEAL: funnel_neal_eval_f32(a, ep, E, st); break;
    case MJHMC_E_FUNNEL_REF: funnel_ref_eval_f32(a, ep, E, st); break;
    default: return fail(MJHMC_ERR_UNSUPPORTED, "energy kind has no evaluation kernel yet");
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// helpers on a sampler
// ---------------------------------------------------------------------------------------------
int ensure_stage(mjhmc_sampler* s, size_t elems) {
  if (s->stage_elems >= elems) return 0;
  if (s->stage) HIPCHK(hipFree(s->stage));
  s->stage = nullptr;
  s->stage_elems = 0;
  HIPCHK(hipMalloc(&s->stage, elems * sizeof(double)));
  s->stage_elems = elems;
  return 0;
}

// host (D, N) float64 -> device particle-major matrix `dst` (stream ordered)
static int upload_matrix(mjhmc_sampler* s, const double* host, void* dst) {
  TRY(ensure_stage(s, (size_t)s->D * s->N));
  HIPCHK(hipMemcpyAsync(s->stage, host, (size_t)s->D * s->N * sizeof(double), hipMemcpyHostToDevice, s->stream));
  dim3 grid((unsigned)((s->N + 31) / 32), (unsigned)((s->D + 31) / 32)), block(32, 8);
  if (s->dtype == MJHMC_F64)
    hipLaunchKernelGGL(to_particle_major<double>, grid, block, 0, s->stream, s->stage, (double*)dst, s->D, s->N,
                       s->sh.pitch);
  else if (s->dtype == MJHMC_BF16)
    hipLaunchKernelGGL(to_particle_major<__bf16>, grid, block, 0, s->stream, s->stage, (__bf16*)dst, s->D, s->N,
                       s->sh.pitch);
  else
    hipLaunchKernelGGL(to_particle_major<float>, grid, block, 0, s->stream, s->stage, (float*)dst, s->D, s->N,
                       s->sh.pitch);
  HIPCHK(hipGetLastError());
  return 0;
}

// Device -> pageable host memory, large blocks (the sample ring, a state matrix).  A plain hipMemcpy to pageable memory
// stages through the runtime's own small pinned buffer on one thread (measured 16-25 GB/s of PCIe Gen5's ~55, less on a
// freshly allocated destination whose pages are still to be faulted in).  Here: two pinned buffers of the sampler; the
// DMA of chunk k + 1 runs while kCopyThreads host threads copy chunk k out of its pinned buffer (they also fault the
// destination pages in, in parallel).  Small blocks take the direct copy.
constexpr size_t kPipeChunk = (size_t)32 << 20;
constexpr int kCopyThreads = 8;

static int ensure_pipe(mjhmc_sampler* s) {
  for (int i = 0; i < 2; ++i) {
    if (!s->pipe_pin[i]) HIPCHK(hipHostMalloc(&s->pipe_pin[i], kPipeChunk, hipHostMallocDefault));
    if (!s->pipe_ev[i]) HIPCHK(hipEventCreateWithFlags(&s->pipe_ev[i], hipEventDisableTiming));
  }
  return 0;
}

int copy_to_host(mjhmc_sampler* s, const void* dev_src, void* host_dst, size_t bytes) {
  if (bytes < 2 * kPipeChunk) {
    HIPCHK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
  }
  TRY(ensure_pipe(s));
  const size_t nchunk = (bytes + kPipeChunk - 1) / kPipeChunk;
  auto issue = [&](size_t k) -> int {
    const size_t off = k * kPipeChunk, len = std::min(kPipeChunk, bytes - off);
    HIPCHK(hipMemcpyAsync(s->pipe_pin[k & 1], (const char*)dev_src + off, len, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipEventRecord(s->pipe_ev[k & 1], s->stream));
    return 0;
  };
  TRY(issue(0));
  for (size_t k = 0; k < nchunk; ++k) {
    if (k + 1 < nchunk) TRY(issue(k + 1));  // its buffer was emptied by the host copy of chunk k - 1 (joined below)
    HIPCHK(hipEventSynchronize(s->pipe_ev[k & 1]));
    const size_t off = k * kPipeChunk, len = std::min(kPipeChunk, bytes - off);
    const char* src = (const char*)s->pipe_pin[k & 1];
    char* dst = (char*)host_dst + off;
    const size_t slice = (len / kCopyThreads + 4095) / 4096 * 4096;
    std::thread th[kCopyThreads];
    int nth = 0;
    for (size_t o = slice; o < len; o += slice) {
      const size_t l = std::min(slice, len - o);
      th[nth++] = std::thread([=] { std::memcpy(dst + o, src + o, l); });
    }
    std::memcpy(dst, src, std::min(slice, len));
    for (int t = 0; t < nth; ++t) th[t].join();
  }
  return 0;
}

// device particle-major rows -> host float64 with strides (see to_dim_major)
int download_cols(mjhmc_sampler* s, const void* src, const int64_t* dev_idx, int64_t ncols, double* host,
                  size_t host_elems, int64_t rs, int64_t cs, int64_t off, bool copy_out) {
  dim3 grid((unsigned)((ncols + 31) / 32), (unsigned)((s->D + 31) / 32)), block(32, 8);
  if (s->dtype == MJHMC_F64)
    hipLaunchKernelGGL(to_dim_major<double>, grid, block, 0, s->stream, (const double*)src, dev_idx, s->stage, s->D,
                       ncols, s->sh.pitch, rs, cs, off);
  else if (s->dtype == MJHMC_BF16 && !s->download_f32)
    hipLaunchKernelGGL(to_dim_major<__bf16>, grid, block, 0, s->stream, (const __bf16*)src, dev_idx, s->stage, s->D,
                       ncols, s->sh.pitch, rs, cs, off);
  else
    hipLaunchKernelGGL(to_dim_major<float>, grid, block, 0, s->stream, (const float*)src, dev_idx, s->stage, s->D,
                       ncols, s->sh.pitch, rs, cs, off);
  HIPCHK(hipGetLastError());
  if (copy_out) TRY(copy_to_host(s, s->stage, host, host_elems * sizeof(double)));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// mjhmc_iterate_download: samples cross PCIe WHILE the sampler iterates.
// HMCBase.sample / ContinuousTimeHMC.sample (markov_jump_hmc.py:150-173, 293-338) append state.copy().X after every
// iteration; here iteration i writes its X into ring slot ring_slot0 + i and, as soon as its kernels are done, a
// worker thread of the call re-tiles that slot on a second stream into the reference's (ndims, particles) layout and
// brings it to the caller's time-major array (pinned double buffer, host copy threads) -- beside iterations
// i + 1, i + 2, ... on the sampler's own streams.  The launch code marks an iteration by recording an event on every
// stream that runs part of it (dl_mark); the worker never reads a slot whose events have not completed.
// ---------------------------------------------------------------------------------------------
struct DlSession {
  std::vector<std::vector<hipEvent_t>> ev;   // [iteration] the events that complete it
  std::atomic<int> marked{0};                // iterations whose events are recorded
  std::atomic<bool> launched{false};         // the iterate call has returned: nothing more will be marked
  int n_iter = 0, ring_slot0 = 0;
  double* host = nullptr;                    // (D, n_total * N) float64, time-major
  int64_t n_total = 0, k0 = 0;
  std::atomic<int> downloaded{0};            // slots [0, downloaded) are in the host array
  std::vector<char> retiled;                 // [iteration] its slot already sits, in the host layout, in staging slot i (see dl_retile_in_stream)
  std::atomic<bool> cancel{false};           // dl_restart: the worker leaves at its next look
  std::atomic<bool> exited{false};           // the worker has left: nothing of the session is touched from its thread any more
  bool off = false;                          // after dl_restart: the launch code marks nothing, the caller downloads when the call is over
  int rc = 0;
  std::string err;
};

// particle-major rows [start, start + ncols) of a state matrix -> columns [start, start + ncols) of the compact (D, N) float64
// matrix `stage`: the reference's layout
static int dl_retile(mjhmc_sampler* s, const void* rows0, int64_t start, int64_t ncols, double* stage, hipStream_t st) {
  const char* src = (const char*)rows0 + (size_t)start * row_bytes(s);
  dim3 grid((unsigned)((ncols + 31) / 32), (unsigned)((s->D + 31) / 32)), block(32, 8);
  if (s->dtype == MJHMC_F64)
    hipLaunchKernelGGL(to_dim_major<double>, grid, block, 0, st, (const double*)src, (const int64_t*)nullptr, stage, s->D, ncols,
                       s->sh.pitch, s->N, (int64_t)1, start);
  else if (s->dtype == MJHMC_BF16)
    hipLaunchKernelGGL(to_dim_major<__bf16>, grid, block, 0, st, (const __bf16*)src, (const int64_t*)nullptr, stage, s->D, ncols,
                       s->sh.pitch, s->N, (int64_t)1, start);
  else
    hipLaunchKernelGGL(to_dim_major<float>, grid, block, 0, st, (const float*)src, (const int64_t*)nullptr, stage, s->D, ncols,
                       s->sh.pitch, s->N, (int64_t)1, start);
  HIPCHK(hipGetLastError());
  return 0;
}

// called by the launch code once the kernels completing iterations [i0, i0 + n) of the call are queued
static int dl_mark(mjhmc_sampler* s, int i0, int n, const hipStream_t* streams, int n_streams) {
  DlSession* d = s->dl;
  if (!d || d->off || i0 < d->marked.load(std::memory_order_acquire)) return 0;   // (a re-run after a failure rewrites marked slots with the same values)
  for (int i = i0; i < i0 + n && i < d->n_iter; ++i) {
    for (int k = 0; k < n_streams; ++k) {
      hipEvent_t e = nullptr;
      if (i == i0) {
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHK(hipEventRecord(e, streams[k]));
      } else {
        e = d->ev[(size_t)i0][(size_t)k];    // iterations of one fused launch complete together
      }
      d->ev[(size_t)i].push_back(e);
    }
  }
  d->marked.store(std::min(i0 + n, d->n_iter), std::memory_order_release);
  return 0;
}

static int dl_download_slot(mjhmc_sampler* s, DlSession* d, int i) {
  const size_t elems = (size_t)s->D * s->N;
  const size_t bytes = elems * sizeof(double), row = (size_t)s->N * sizeof(double);
  const void* src = (const char*)s->ring + (size_t)(d->ring_slot0 + i) * mat_bytes(s);
  const bool retiled = d->retiled[(size_t)i] != 0;
  const double* stage = retiled ? s->dl_stage + (size_t)i * elems : s->dl_stage;
  if (!retiled) TRY(dl_retile(s, src, 0, s->N, s->dl_stage, s->dl_stream));
  // compact (D, N) staging -> row d of the slot lands at host[(d * n_total + k0 + i) * N]; chunks through the pinned pair
  char* host = (char*)d->host;
  const size_t dst_row = (size_t)d->n_total * row, dst_off = (size_t)(d->k0 + i) * row;
  const size_t nchunk = (bytes + kPipeChunk - 1) / kPipeChunk;
  auto issue = [&](size_t k) -> int {
    const size_t off = k * kPipeChunk, len = std::min(kPipeChunk, bytes - off);
    HIPCHK(hipMemcpyAsync(s->dl_pin[k & 1], (const char*)stage + off, len, hipMemcpyDeviceToHost, s->dl_stream));
    HIPCHK(hipEventRecord(s->dl_ev[k & 1], s->dl_stream));
    return 0;
  };
  TRY(issue(0));
  for (size_t k = 0; k < nchunk; ++k) {
    if (k + 1 < nchunk) TRY(issue(k + 1));
    HIPCHK(hipEventSynchronize(s->dl_ev[k & 1]));
    const size_t off = k * kPipeChunk, len = std::min(kPipeChunk, bytes - off);
    const char* pin = (const char*)s->dl_pin[k & 1];
    // the chunk's bytes [off, off + len) of the compact matrix, row by row, over the copy threads
    auto scatter = [=](size_t lo, size_t hi) {   // byte range of the chunk
      size_t o = lo;
      while (o < hi) {
        const size_t g = off + o, dd = g / row, within = g % row;
        const size_t n = std::min(row - within, hi - o);
        std::memcpy(host + dd * dst_row + dst_off + within, pin + o, n);
        o += n;
      }
    };
    const size_t slice = (len / kCopyThreads + 4095) / 4096 * 4096;
    std::thread th[kCopyThreads];
    int nth = 0;
    for (size_t o = slice; o < len; o += slice) th[nth++] = std::thread(scatter, o, std::min(o + slice, len));
    scatter(0, std::min(slice, len));
    for (int t = 0; t < nth; ++t) th[t].join();
  }
  return 0;
}

static void dl_worker_body(mjhmc_sampler* s, DlSession* d) {
  if (hipSetDevice(s->ctx->device) != hipSuccess) {
    d->rc = MJHMC_ERR_HIP;
    d->err = "hipSetDevice failed in the download thread";
    return;
  }
  for (int i = 0; i < d->n_iter; ++i) {
    while (d->marked.load(std::memory_order_acquire) <= i && !d->launched.load(std::memory_order_acquire) &&
           !d->cancel.load(std::memory_order_acquire))
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (d->cancel.load(std::memory_order_acquire)) return;
    if (d->marked.load(std::memory_order_acquire) <= i) break;   // a path that marks nothing: the caller downloads afterwards
    for (hipEvent_t e : d->ev[(size_t)i])
      if (hipStreamWaitEvent(s->dl_stream, e, 0) != hipSuccess) {
        d->rc = MJHMC_ERR_HIP;
        d->err = "hipStreamWaitEvent failed in the download thread";
        return;
      }
    const int rc = dl_download_slot(s, d, i);
    if (rc) {
      d->rc = rc;
      d->err = g_err;   // (thread-local: hand the message to the calling thread)
      return;
    }
    if (d->cancel.load(std::memory_order_acquire)) return;   // (the slot it has just brought over is about to be rewritten)
    d->downloaded.store(i + 1, std::memory_order_release);
  }
}

static void dl_worker(mjhmc_sampler* s, DlSession* d) {
  dl_worker_body(s, d);
  d->exited.store(true, std::memory_order_release);
}

// the events of a session, each destroyed once (the iterations of a fused launch share theirs)
static void dl_destroy_events(DlSession* d) {
  for (size_t i = 0; i < d->ev.size(); ++i)
    for (hipEvent_t e : d->ev[i]) {
      bool first = true;
      for (size_t j = 0; j < i && first; ++j)
        for (hipEvent_t f : d->ev[j]) first = first && f != e;
      if (first) (void)hipEventDestroy(e);
    }
  for (auto& v : d->ev) v.clear();
}

// A call that is about to be run AGAIN from its first iteration (iterate_t: a non-finite rate while the parts of a dense
// batch ran freely -- the state is put back and the call re-run on one stream) while samples are being downloaded: what
// the first run marked is void.  A lagging part returned early in the failing iteration and left its columns of the
// ring slots and of the staging slots stale, and the re-run rewrites every slot: the worker must neither have copied nor
// copy any of them.  The worker is told to leave and waited for, its stream drained, the session emptied and switched
// off -- the re-run marks nothing, and mjhmc_iterate_download brings slots [0, done) over when the call has returned.
static int dl_restart(mjhmc_sampler* s) {
  DlSession* d = s->dl;
  if (!d) return 0;
  d->cancel.store(true, std::memory_order_release);
  while (!d->exited.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(50));
  if (s->dl_stream) HIPCHK(hipStreamSynchronize(s->dl_stream));
  dl_destroy_events(d);
  d->marked.store(0, std::memory_order_release);
  d->downloaded.store(0, std::memory_order_release);
  std::fill(d->retiled.begin(), d->retiled.end(), 0);
  d->off = true;
  return 0;
}

template <typename T>
static int run_eval_t(mjhmc_sampler* s, const void* X, void* Gout, void* Eout, const void* V, void* Vgen, void* EVout) {
  EvalArgs<T> a;
  a.X = (const T*)X;
  a.G = (T*)Gout;
  a.E = (T*)Eout;
  a.EV = (T*)EVout;
  a.V = (const T*)V;
  a.V_out = (T*)Vgen;
  a.N = s->N;
  a.first_pid = s->first_pid;
  a.D = s->D;
  a.pitch = s->sh.pitch;
  a.CH = s->sh.CH;
  a.logG = s->sh.logG;
  a.key = RngKey{(uint32_t)(s->seed & 0xFFFFFFFFu), (uint32_t)(s->seed >> 32), 0u, 0u};
  if constexpr (sizeof(T) == 8) {
    if (s->en->is_user()) return user_launch_eval(s->en, a, s->stream);
  }
  TRY(dispatch_eval<T>(s->en->ep.kind, a, s->en->ep, s->sh.E, s->stream));
  HIPCHK(hipGetLastError());
  return 0;
}

static int run_eval_pot(mjhmc_sampler* s, const void* X, void* Gout, void* Eout, const void* V, void* Vgen,
                        void* EVout) {
  PotEvalArgs a;
  a.X = (const float*)X;
  a.G = (float*)Gout;
  a.E = (float*)Eout;
  a.EV = (float*)EVout;
  a.V = (const float*)V;
  a.V_gen = (float*)Vgen;
  a.N = s->N;
  a.ntiles = s->Npad / 32;
  a.first_pid = s->first_pid;
  a.D = s->D;
  a.key = RngKey{(uint32_t)(s->seed & 0xFFFFFFFFu), (uint32_t)(s->seed >> 32), 0u, 0u};
  pot_launch_eval(a, s->en->pot_model(), s->stream);
  HIPCHK(hipGetLastError());
  return 0;
}

template <typename ST>
static int run_eval_sic_t(mjhmc_sampler* s, const void* X, void* Gout, void* Eout, const void* V, void* Vgen,
                          void* EVout) {
  SicEvalArgsT<ST> a;
  a.X = (const ST*)X;
  a.G = (float*)Gout;  // float32 [Npad][1024]
  a.E = (float*)Eout;
  a.EV = (float*)EVout;
  a.V = (const ST*)V;
  a.V_gen = (ST*)Vgen;
  a.N = s->N;
  const int ppt = sic_particles_per_tile(s->en->sic_P);
  a.ntiles = (s->N + ppt - 1) / ppt;
  a.first_pid = s->first_pid;
  a.key = RngKey{(uint32_t)(s->seed & 0xFFFFFFFFu), (uint32_t)(s->seed >> 32), 0u, 0u};
  sic_launch_eval(a, s->en->sic_model(), s->stream);
  HIPCHK(hipGetLastError());
  return 0;
}
static int run_eval_sic(mjhmc_sampler* s, const void* X, void* Gout, void* Eout, const void* V, void* Vgen, void* EVout) {
  return s->dtype == MJHMC_BF16 ? run_eval_sic_t<__bf16>(s, X, Gout, Eout, V, Vgen, EVout)
                                : run_eval_sic_t<float>(s, X, Gout, Eout, V, Vgen, EVout);
}

static int run_eval(mjhmc_sampler* s, const void* X, void* Gout, void* Eout, const void* V, void* Vgen, void* EVout) {
  // host energies (E and dE/dX are the caller's, mjhmc_host_set_energy) and wide rows (multi-pass device kernels)
  if (s->en->is_host() || s->sh.wide) return wide_run_eval(s, X, Gout, Eout, V, Vgen, EVout);
  if (s->en->is_pot()) return run_eval_pot(s, X, Gout, Eout, V, Vgen, EVout);
  if (s->en->is_sic()) return run_eval_sic(s, X, Gout, Eout, V, Vgen, EVout);
  return s->dtype == MJHMC_F64 ? run_eval_t<double>(s, X, Gout, Eout, V, Vgen, EVout)
                               : run_eval_t<float>(s, X, Gout, Eout, V, Vgen, EVout);
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

const char* mjhmc_last_error(void) { return g_err.c_str(); }
int mjhmc_abi_version(void) { return MJHMC_ABI_VERSION; }

int mjhmc_ctx_create(int device, mjhmc_ctx** out) {
  if (!out) return fail(MJHMC_ERR_INVALID, "out is NULL");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(MJHMC_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= n) return fail(MJHMC_ERR_INVALID, "device index out of range");
  HIPCHK(hipSetDevice(device));
  mjhmc_ctx* c = new mjhmc_ctx();
  c->device = device;
  HIPCHK(hipGetDeviceProperties(&c->prop, device));
  *out = c;
  return 0;
}

int mjhmc_ctx_destroy(mjhmc_ctx* ctx) {
  delete ctx;
  return 0;
}

int mjhmc_ctx_info(mjhmc_ctx* ctx, char* name, size_t name_cap, int* n_cu, uint64_t* hbm_bytes) {
  if (!ctx) return fail(MJHMC_ERR_INVALID, "ctx is NULL");
  if (name && name_cap) {
    // ... and its PCI address, e.g. "[0000:75:00.0]": how a caller finds the device's sysfs / SMI entry
    std::snprintf(name, name_cap, "%s (%s) [%04x:%02x:%02x.0]", ctx->prop.name, ctx->prop.gcnArchName, ctx->prop.pciDomainID,
                  ctx->prop.pciBusID, ctx->prop.pciDeviceID);
  }
  if (n_cu) *n_cu = ctx->prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (uint64_t)ctx->prop.totalGlobalMem;
  return 0;
}

int mjhmc_energy_create(mjhmc_ctx* ctx, int kind, int ndims, const double* params, size_t nparams,
                        mjhmc_energy** out) {
  if (!ctx || !out) return fail(MJHMC_ERR_INVALID, "ctx/out is NULL");
  if (ndims < 1) return fail(MJHMC_ERR_INVALID, "ndims must be >= 1");
  if (nparams && !params) return fail(MJHMC_ERR_INVALID, "params is NULL");
  HIPCHK(hipSetDevice(ctx->device));
  mjhmc_energy* e = new mjhmc_energy();
  e->ctx = ctx;
  e->params.assign(params, params + nparams);
  std::memset(&e->ep, 0, sizeof(e->ep));
  e->ep.kind = kind;
  e->ep.ndims = ndims;
  auto need = [&](size_t n) { return nparams == n; };
  int rc = 0;
  switch (kind) {
    case MJHMC_E_ISO_GAUSS:
      if (!need(1) || !(params[0] > 0)) rc = fail(MJHMC_ERR_INVALID, "ISO_GAUSS expects {sigma > 0}");
      else e->ep.p[0] = params[0];
      break;
    case MJHMC_E_ROUGH_WELL:
      if (!need(2)) rc = fail(MJHMC_ERR_INVALID, "ROUGH_WELL expects {scale1, scale2}");
      else {
        e->ep.p[0] = params[0];
        e->ep.p[1] = params[1];
      }
      break;
    case MJHMC_E_MM_GAUSS:
    case MJHMC_E_FUNNEL_NEAL:
    case MJHMC_E_FUNNEL_REF:
      if (!need(1)) rc = fail(MJHMC_ERR_INVALID, "expects one scalar parameter");
      else e->ep.p[0] = params[0];
      break;
    case MJHMC_E_DIAG_GAUSS: {
      if (!need((size_t)ndims)) {
        rc = fail(MJHMC_ERR_INVALID, "DIAG_GAUSS expects ndims diagonal entries");
        break;
      }
      const size_t npad = std::max<size_t>(kParamPad, (size_t)ndims);  // (more than kParamPad dims: the multi-pass path)
      std::vector<double> h64(npad, 0.0);
      std::vector<float> h32(npad, 0.f);
      for (int i = 0; i < ndims; ++i) {
        h64[i] = params[i];
        h32[i] = (float)params[i];
      }
      if (hipMalloc(&e->dev64, npad * sizeof(double)) != hipSuccess ||
          hipMalloc(&e->dev32, npad * sizeof(float)) != hipSuccess ||
          hipMemcpy(e->dev64, h64.data(), npad * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
          hipMemcpy(e->dev32, h32.data(), npad * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        rc = fail(MJHMC_ERR_HIP, "allocating DIAG_GAUSS parameters failed");
      e->ep.dev_f64 = e->dev64;
      e->ep.dev_f32 = e->dev32;
      break;
    }
    case MJHMC_E_PRODUCT_OF_T: {
      // params = {nbasis, W[D*nbasis] row-major, nu[nbasis], b[nbasis]}  (distributions.py:379-406)
      const int K = nparams ? (int)params[0] : 0;
      if (K != ndims || nparams != (size_t)1 + (size_t)ndims * K + 2 * (size_t)K) {
        rc = fail(MJHMC_ERR_INVALID, "PRODUCT_OF_T expects {nbasis == ndims, W[D*K], nu[K], b[K]}");
        break;
      }
      const double* W = params + 1;
      const double* nu = W + (size_t)ndims * K;
      const double* b = nu + K;
      // up to 512 dims: the register-resident tile kernels; beyond: 512 x 512 blocks for the multi-pass path (pot_big_eval)
      const int DIM = ndims <= 128 ? 128 : (ndims <= 256 ? 256 : (ndims + 511) / 512 * 512);
      const int nb = DIM / 512;   // (>= 2: blocked storage)
      e->pot_dim = DIM;
      const size_t M = (size_t)DIM * DIM;
      std::vector<float> w1(M, 0.f), w2t(M, 0.f), cb(DIM, 0.f), al(DIM, 0.f);
      for (int j = 0; j < K; ++j) {
        // parameters are float32 in the reference (theano.shared(np.array(.., dtype='float32')), :398-406)
        const double nuj = (double)(float)nu[j];
        cb[j] = (float)((double)(float)b[j] / nuj);
        al[j] = (float)((nuj + 1.0) / 2.0);
        for (int d = 0; d < ndims; ++d) {
          const double wdj = (double)(float)W[(size_t)d * K + j];
          if (nb < 2) {
            w1[(size_t)d * DIM + j] = (float)(wdj / nuj);
            w2t[(size_t)j * DIM + d] = (float)(wdj * (nuj + 1.0) / nuj);
          } else {   // W1b[db][jb][d % 512][j % 512], W2Tb[jb][db][j % 512][d % 512]
            const size_t db = (size_t)d / 512, jb = (size_t)j / 512, dl = (size_t)d % 512, jl = (size_t)j % 512;
            w1[((db * nb + jb) * 512 + dl) * 512 + jl] = (float)(wdj / nuj);
            w2t[((jb * nb + db) * 512 + jl) * 512 + dl] = (float)(wdj * (nuj + 1.0) / nuj);
          }
        }
      }
      const void* src[4] = {w1.data(), w2t.data(), cb.data(), al.data()};
      const size_t bytes[4] = {M * 4, M * 4, (size_t)DIM * 4, (size_t)DIM * 4};
      for (int i = 0; i < 4 && !rc; ++i) {
        if (hipMalloc((void**)&e->pot[i], bytes[i]) != hipSuccess ||
            hipMemcpy(e->pot[i], src[i], bytes[i], hipMemcpyHostToDevice) != hipSuccess)
          rc = fail(MJHMC_ERR_HIP, "allocating PRODUCT_OF_T parameters failed");
      }
      break;
    }
    case MJHMC_E_SPARSE_CODE: {
      // params = {n_patches, img, n_coeffs, lambda, cauchy, B[img*n_coeffs] (img, n_coeffs) row-major, Y[n_patches*img]}
      if (nparams < 5) {
        rc = fail(MJHMC_ERR_INVALID, "SPARSE_CODE expects {n_patches, img, n_coeffs, lambda, cauchy, B, Y}");
        break;
      }
      const int P = (int)params[0], I = (int)params[1], C = (int)params[2];
      if (P < 1 || P > 32 || I != kSicImg || !sic_coeffs_supported(C) || ndims != P * C) {
        rc = fail(MJHMC_ERR_UNSUPPORTED,
                  "SPARSE_CODE device kernel is built for img_size=256, n_coeffs=1024 or 512, 1 <= n_patches <= 32");
        break;
      }
      e->sic_P = P;
      e->sic_nc = C;
      if (nparams != (size_t)5 + (size_t)I * C + (size_t)P * I) {
        rc = fail(MJHMC_ERR_INVALID, "SPARSE_CODE parameter vector has the wrong length");
        break;
      }
      e->sic_lambda = (float)params[3];
      e->sic_cauchy = params[4] != 0.0 ? 1 : 0;
      const double* B = params + 5;
      const double* Y = B + (size_t)I * C;
      auto bf16_of = [](double v) -> uint16_t {  // round-to-nearest-even float32 -> bfloat16
        float f = (float)v;
        uint32_t u;
        std::memcpy(&u, &f, 4);
        if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
        u += 0x7FFFu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
      };
      const int NB = C / 256;  // 32-row blocks per wave (8 waves share the C coefficient rows)
      std::vector<uint16_t> a1((size_t)(C / 16) * 2 * I * 8), a2((size_t)16 * 2 * C * 8);
      for (int ks = 0; ks < C / 16; ++ks)
        for (int h = 0; h < 2; ++h)
          for (int i = 0; i < I; ++i)
            for (int j = 0; j < 8; ++j) {
              const int ws = ks / (2 * NB), b = (ks >> 1) % NB, sx = ks & 1;
              const int c = 32 * NB * ws + 32 * b + 16 * sx + 8 * (j >> 2) + 4 * h + (j & 3);
              a1[(((size_t)ks * 2 + h) * I + i) * 8 + j] = bf16_of(B[(size_t)i * C + c]);
            }
      for (int ks = 0; ks < 16; ++ks)
        for (int h = 0; h < 2; ++h)
          for (int c = 0; c < C; ++c)
            for (int j = 0; j < 8; ++j) {
              const int ws = ks >> 1, sx = ks & 1;
              const int i = 32 * ws + 16 * sx + 8 * (j >> 2) + 4 * h + (j & 3);
              a2[(((size_t)ks * 2 + h) * C + c) * 8 + j] = bf16_of(B[(size_t)i * C + c]);
            }
      std::vector<float> yv((size_t)P * I);
      for (size_t i = 0; i < yv.size(); ++i) yv[i] = (float)Y[i];
      const void* src[3] = {a1.data(), a2.data(), yv.data()};
      const size_t bytes[3] = {a1.size() * 2, a2.size() * 2, yv.size() * 4};
      // the two fragment orders of the dictionary are kept in kSicCopies identical copies (workgroups of an XCD
      // alternate between them): every CU streams the same 1 MB per leapfrog step out of its XCD's L2, and spreading
      // that over two sets of lines was measured faster than all CUs hitting one
      e->sic_copies = kSicCopies;
      if (const char* cp = test_env("MJHMC_SIC_COPIES")) e->sic_copies = std::max(1, std::min(4, std::atoi(cp)));
      for (int i = 0; i < 3 && !rc; ++i) {
        const int reps = i < 2 ? e->sic_copies : 1;
        if (hipMalloc(&e->sic[i], bytes[i] * reps) != hipSuccess) rc = fail(MJHMC_ERR_HIP, "allocating SPARSE_CODE parameters failed");
        for (int r = 0; r < reps && !rc; ++r)
          if (hipMemcpy((char*)e->sic[i] + (size_t)r * bytes[i], src[i], bytes[i], hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(MJHMC_ERR_HIP, "uploading SPARSE_CODE parameters failed");
      }
      break;
    }
    case MJHMC_E_HOST:  // no device form: the caller evaluates E and dE/dX (host_energy.hip)
      if (nparams) rc = fail(MJHMC_ERR_INVALID, "HOST takes no parameters");
      break;
    default:
      rc = fail(MJHMC_ERR_UNSUPPORTED, "energy kind not implemented in this build");
  }
  if (rc) {
    mjhmc_energy_destroy(e);
    return rc;
  }
  *out = e;
  return 0;
}

int mjhmc_energy_create_expr_coupled(mjhmc_ctx* ctx, int ndims, const char* stat_exprs, const char* energy_expr,
                                     const char* energy0_expr, const char* grad_expr, const double* params, size_t nparams,
                                     const char* include_dir, mjhmc_energy** out) {
  if (!ctx || !out || !energy_expr || !grad_expr || !include_dir) return fail(MJHMC_ERR_INVALID, "NULL argument");
  if (ndims < 1) return fail(MJHMC_ERR_INVALID, "ndims must be >= 1");
  if (nparams && !params) return fail(MJHMC_ERR_INVALID, "params is NULL");
  HIPCHK(hipSetDevice(ctx->device));
  Shape sh;
  TRY(pick_shape(ndims, MJHMC_F64, &sh));
  mjhmc_energy* e = new mjhmc_energy();
  e->ctx = ctx;
  if (nparams) e->params.assign(params, params + nparams);
  std::memset(&e->ep, 0, sizeof(e->ep));
  e->ep.kind = MJHMC_E_USER_EXPR;
  e->ep.ndims = ndims;
  const int rc = user_energy_build(e, energy_expr, grad_expr, stat_exprs, energy0_expr, include_dir, params, nparams, sh.E);
  if (rc) {
    std::string keep = g_err;
    mjhmc_energy_destroy(e);
    g_err = keep;
    return rc;
  }
  *out = e;
  return 0;
}

int mjhmc_energy_create_expr(mjhmc_ctx* ctx, int ndims, const char* energy_expr, const char* grad_expr,
                             const double* params, size_t nparams, const char* include_dir, mjhmc_energy** out) {
  return mjhmc_energy_create_expr_coupled(ctx, ndims, nullptr, energy_expr, nullptr, grad_expr, params, nparams, include_dir, out);
}

int mjhmc_expr_check_coupled(int ndims, const char* stat_exprs, const char* energy_expr, const char* energy0_expr,
                             const char* grad_expr, const char* include_dir) {
  if (!energy_expr || !grad_expr || !include_dir) return fail(MJHMC_ERR_INVALID, "NULL argument");
  if (ndims < 1) return fail(MJHMC_ERR_INVALID, "ndims must be >= 1");
  Shape sh;
  TRY(pick_shape(ndims, MJHMC_F64, &sh));
  std::vector<char> code;
  std::vector<std::string> lowered;
  std::string err;
  const int rc = user_expr_compile(user_expr_source(energy_expr, grad_expr, stat_exprs ? stat_exprs : "", energy0_expr ? energy0_expr : ""),
                                   include_dir, sh.E, &code, &lowered, &err);
  return rc ? fail(rc, err) : 0;
}

int mjhmc_expr_check(int ndims, const char* energy_expr, const char* grad_expr, const char* include_dir) {
  return mjhmc_expr_check_coupled(ndims, nullptr, energy_expr, nullptr, grad_expr, include_dir);
}

int mjhmc_energy_destroy(mjhmc_energy* e) {
  if (!e) return 0;
  user_energy_free(e);
  if (e->dev64) (void)hipFree(e->dev64);
  if (e->dev32) (void)hipFree(e->dev32);
  for (float* q : e->pot)
    if (q) (void)hipFree(q);
  for (void* q : e->sic)
    if (q) (void)hipFree(q);
  delete e;
  return 0;
}

int mjhmc_sampler_destroy(mjhmc_sampler* s) {
  if (!s) return 0;
  (void)hipSetDevice(s->ctx->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  void* ptrs[] = {s->flf_list, s->call_block, s->Hpre, s->Hwork, s->cold_list, s->pot64_scratch, s->Hspec[0], s->Hspec[1], s->Hspec_dump, s->Gbuf[0], s->Gbuf[1], s->Xbuf[0], s->Xbuf[1], s->Vbuf[0],  s->Vbuf[1], s->EX[0],     s->EX[1],  s->EV[0],
                  s->EV[1],   s->Hflf[0], s->Hflf[1],  s->dwell,  s->dwell_scratch,  s->trans,
                  s->ring,     s->dwell_ring, s->stage,  s->noise,  s->rexp,
                  s->runif,   s->scratch,  s->ck[0],    s->ck[1],    s->ck[2],   s->ck[3],  s->ck[4],
                  s->ck[5],   s->ck[6]};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (auto& e : s->ev_total)
    if (e) (void)hipEventDestroy(e);
  for (auto& e : s->ev_k) (void)hipEventDestroy(e);
  for (auto& e : s->part_events) (void)hipEventDestroy(e);
  for (auto& q : s->part_streams) (void)hipStreamDestroy(q);
  if (s->ev_join) (void)hipEventDestroy(s->ev_join);
  if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
  for (void* q : s->ick)
    if (q) (void)hipFree(q);
  if (s->h_pin) (void)hipHostFree(s->h_pin);
  if (s->dl_stage) (void)hipFree(s->dl_stage);
  for (int i = 0; i < 2; ++i) {
    if (s->dl_pin[i]) (void)hipHostFree(s->dl_pin[i]);
    if (s->dl_ev[i]) (void)hipEventDestroy(s->dl_ev[i]);
  }
  if (s->dl_stream) (void)hipStreamDestroy(s->dl_stream);
  for (int i = 0; i < 2; ++i) {
    if (s->pipe_pin[i]) (void)hipHostFree(s->pipe_pin[i]);
    if (s->pipe_ev[i]) (void)hipEventDestroy(s->pipe_ev[i]);
  }
  host_traj_free(s);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
  return 0;
}

int mjhmc_sampler_create(mjhmc_ctx* ctx, mjhmc_energy* e, int64_t nparticles, int64_t first_particle_id, int dtype,
                         const double* Xinit, const double* Vinit, uint64_t seed, int mode, mjhmc_sampler** out) {
  if (!ctx || !e || !out || !Xinit) return fail(MJHMC_ERR_INVALID, "NULL argument");
  if (nparticles < 1) return fail(MJHMC_ERR_INVALID, "nparticles must be >= 1");
  if (dtype != MJHMC_F64 && dtype != MJHMC_F32 && dtype != MJHMC_BF16)
    return fail(MJHMC_ERR_INVALID, "dtype must be F64, F32 or BF16");
  if (dtype == MJHMC_BF16 && !e->is_sic()) return fail(MJHMC_ERR_UNSUPPORTED, "BF16 state: SPARSE_CODE only");
  if (e->is_sic() && dtype == MJHMC_F64)
    return fail(MJHMC_ERR_UNSUPPORTED, "SPARSE_CODE runs with BF16 state (the benchmark's) or F32 state (the reference's)");
  if (e->is_user() && dtype != MJHMC_F64) return fail(MJHMC_ERR_UNSUPPORTED, "user-expression energies run in float64");
  if (e->is_host() && dtype != MJHMC_F64) return fail(MJHMC_ERR_UNSUPPORTED, "host-evaluated energies run in float64");
  if (mode < MJHMC_MODE_MJHMC || mode > MJHMC_MODE_CTHMC) return fail(MJHMC_ERR_INVALID, "unknown sampler mode");
  if (first_particle_id < 0 || first_particle_id + nparticles > 0xFFFFFFFFLL)
    return fail(MJHMC_ERR_INVALID, "global particle ids must fit 32 bits");
  HIPCHK(hipSetDevice(ctx->device));
  mjhmc_sampler* s = new mjhmc_sampler();
  s->ctx = ctx;
  s->en = e;
  s->N = nparticles;
  s->Npad = (nparticles + 63) / 64 * 64;
  s->first_pid = first_particle_id;
  s->D = e->ep.ndims;
  s->dtype = dtype;
  s->mode = mode;
  s->seed = seed;
  int rc = shape_for(e, &s->dtype, &s->sh);
  if (rc) {
    delete s;
    return rc;
  }
  dtype = s->dtype;   // (the storage dtype)
  auto build = [&]() -> int {
    HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    const size_t mb = mat_bytes(s);
    for (int i = 0; i < 2; ++i) {
      HIPCHK(hipMalloc(&s->Xbuf[i], mb));
      HIPCHK(hipMalloc(&s->Vbuf[i], mb));
      HIPCHK(hipMemsetAsync(s->Xbuf[i], 0, mb, s->stream));
      HIPCHK(hipMemsetAsync(s->Vbuf[i], 0, mb, s->stream));
      if (e->is_pot() || ((e->is_host() || s->sh.wide) && i == 0)) {  // these keep dE/dX as part of the state, like HMCState.dEdX
        HIPCHK(hipMalloc(&s->Gbuf[i], mb));
        HIPCHK(hipMemsetAsync(s->Gbuf[i], 0, mb, s->stream));
      }
      if (e->is_dense()) {
        if (!s->Hwork) {
          HIPCHK(hipMalloc((void**)&s->Hwork, s->Npad * ssize(s)));  // float32, or float64 for ProductOfT's float64 state
          HIPCHK(hipMemsetAsync(s->Hwork, 0, s->Npad * ssize(s), s->stream));   // (padding rows read it and ignore it)
          HIPCHK(hipMalloc((void**)&s->cold_list, (2 * s->Npad + 3 * kMaxDenseParts) * sizeof(int)));  // two lists (iterations alternate) + [part][3] counters
          if (e->is_pot() && dtype == MJHMC_F64 && !e->pot_big())   // working rows of the tile kernel, one set per concurrent launch
            HIPCHK(hipMalloc((void**)&s->pot64_scratch, (size_t)kMaxDenseParts * pot64_scratch_workgroups() * 2 * 32 * row_bytes(s)));
        }
        HIPCHK(hipMalloc(&s->Hspec[i], s->Npad * ssize(s)));
        HIPCHK(hipMemsetAsync(s->Hspec[i], 0xFF, s->Npad * ssize(s), s->stream));   // NaN: nothing handed on
      }
      HIPCHK(hipMalloc(&s->EX[i], s->Npad * ssize(s)));
      HIPCHK(hipMalloc(&s->EV[i], s->Npad * ssize(s)));
      HIPCHK(hipMalloc(&s->Hflf[i], s->Npad * ssize(s)));
      HIPCHK(hipMemsetAsync(s->EX[i], 0, s->Npad * ssize(s), s->stream));
      HIPCHK(hipMemsetAsync(s->EV[i], 0, s->Npad * ssize(s), s->stream));
      HIPCHK(hipMemsetAsync(s->Hflf[i], 0xFF, s->Npad * ssize(s), s->stream));  // all-ones = NaN = cold
    }
    HIPCHK(hipMalloc((void**)&s->dwell, s->Npad * sizeof(double)));
    HIPCHK(hipMemsetAsync(s->dwell, 0, s->Npad * sizeof(double), s->stream));
    HIPCHK(hipMalloc((void**)&s->dwell_scratch, s->Npad * sizeof(double)));
    HIPCHK(hipMalloc((void**)&s->trans, s->Npad));
    HIPCHK(hipMemsetAsync(s->trans, 0, s->Npad, s->stream));
    TRY(ensure_call_block(s, 1024));   // everything mjhmc_iterate needs is created here, so a timed call never allocates
    HIPCHK(hipEventCreate(&s->ev_total[0]));
    HIPCHK(hipEventCreate(&s->ev_total[1]));
    // a state matrix big enough for the pipelined host copies: their pinned buffers now, not inside the first read
    if ((size_t)s->D * s->N * sizeof(double) >= 2 * kPipeChunk) TRY(ensure_pipe(s));
    s->Xcur = s->Xbuf[0];
    TRY(upload_matrix(s, Xinit, s->Xcur));
    if (s->sh.round32) TRY(round_rows32(s, s->Xcur));
    if (Vinit) {
      TRY(upload_matrix(s, Vinit, s->Vbuf[0]));
      if (s->sh.round32) TRY(round_rows32(s, s->Vbuf[0]));
      TRY(run_eval(s, s->Xcur, s->Gbuf[0], s->EX[0], s->Vbuf[0], nullptr, s->EV[0]));
    } else {
      TRY(run_eval(s, s->Xcur, s->Gbuf[0], s->EX[0], nullptr, s->Vbuf[0], s->EV[0]));
      if (s->sh.round32) {   // the generated momentum is stored in float32 too: its kinetic energy is that of what is stored
        TRY(round_rows32(s, s->Vbuf[0]));
        TRY(run_eval(s, nullptr, nullptr, nullptr, s->Vbuf[0], nullptr, s->EV[0]));
      }
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    return 0;
  };
  rc = build();
  if (rc) {
    std::string keep = g_err;
    mjhmc_sampler_destroy(s);
    g_err = keep;
    return rc;
  }
  *out = s;
  return 0;
}

// An F-mover's H(L proposal) stands in for the H of its inverse-L proposal in the next iteration (dense_pot.hip).  That
// holds for the state and the trajectory it was computed with: anything that changes either -- the step size or count,
// a state or cache write from the host, reset_flf_cache, a restore -- drops it (all NaN: every cold particle integrates).
static int drop_spec(mjhmc_sampler* s) {
  if (!s->Hspec[0]) return 0;
  HIPCHK(hipSetDevice(s->ctx->device));
  HIPCHK(hipMemsetAsync(s->Hspec[s->scur], 0xFF, (size_t)s->Npad * ssize(s), s->stream));
  return 0;
}

int mjhmc_set_hparams(mjhmc_sampler* s, double epsilon, int num_leapfrog_steps, double p_r, double beta,
                      double p_flip) {
  if (!s) return fail(MJHMC_ERR_INVALID, "sampler is NULL");
  if (num_leapfrog_steps < 0) return fail(MJHMC_ERR_INVALID, "num_leapfrog_steps must be >= 0");
  if (epsilon != s->eps || num_leapfrog_steps != s->L) TRY(drop_spec(s));
  s->eps = epsilon;
  s->L = num_leapfrog_steps;
  s->p_r = p_r;
  s->beta = beta;
  s->p_flip = p_flip;
  return 0;
}

int mjhmc_checkpoint(mjhmc_sampler* s) {
  if (!s) return fail(MJHMC_ERR_INVALID, "sampler is NULL");
  HIPCHK(hipSetDevice(s->ctx->device));
  const size_t mb = mat_bytes(s), vb = (size_t)s->Npad * ssize(s), db = (size_t)s->Npad * sizeof(double);
  // ProductOfT keeps dE/dX of the current state (HMCState.dEdX) and the jump kernel does not recompute it: it is
  // part of the state a rollback must put back
  const int nck = (s->en->is_pot() || s->en->is_host() || s->sh.wide) ? 7 : 6;
  const size_t sizes[7] = {mb, mb, vb, vb, vb, db, mb};
  const void* src[7] = {s->Xcur, s->Vbuf[s->vcur], s->EX[s->scur], s->EV[s->scur], s->Hflf[s->scur], s->dwell,
                        s->Gbuf[s->vcur]};
  for (int i = 0; i < nck; ++i) {
    if (!s->ck[i]) HIPCHK(hipMalloc(&s->ck[i], sizes[i]));
    HIPCHK(hipMemcpyAsync(s->ck[i], src[i], sizes[i], hipMemcpyDeviceToDevice, s->stream));
  }
  s->ck_tick = s->tick;
  s->ck_valid = true;
  return 0;
}

int mjhmc_restore(mjhmc_sampler* s) {
  if (!s) return fail(MJHMC_ERR_INVALID, "sampler is NULL");
  if (!s->ck_valid) return fail(MJHMC_ERR_INVALID, "no checkpoint taken");
  HIPCHK(hipSetDevice(s->ctx->device));
  const size_t mb = mat_bytes(s), vb = (size_t)s->Npad * ssize(s), db = (size_t)s->Npad * sizeof(double);
  const int nck = (s->en->is_pot() || s->en->is_host() || s->sh.wide) ? 7 : 6;
  const size_t sizes[7] = {mb, mb, vb, vb, vb, db, mb};
  s->Xcur = s->Xbuf[0];
  void* dst[7] = {s->Xcur, s->Vbuf[s->vcur], s->EX[s->scur], s->EV[s->scur], s->Hflf[s->scur], s->dwell,
                  s->Gbuf[s->vcur]};
  for (int i = 0; i < nck; ++i) HIPCHK(hipMemcpyAsync(dst[i], s->ck[i], sizes[i], hipMemcpyDeviceToDevice, s->stream));
  TRY(drop_spec(s));
  s->list_valid = false;
  s->tick = s->ck_tick;
  s->undo_valid = false;
  HIPCHK(hipStreamSynchronize(s->stream));
  return 0;
}

int mjhmc_rollback(mjhmc_sampler* s) {
  if (!s) return fail(MJHMC_ERR_INVALID, "sampler is NULL");
  if (!s->undo_valid) return fail(MJHMC_ERR_INVALID, "nothing to roll back: the last call was not a committed single iteration");
  // an iteration reads one buffer parity and writes the other: flipping the parities back IS the pre-move state
  // (X, V, EX, EV, H_flf and, for ProductOfT, dE/dX); dwell / trans hold the rolled-back attempt's values until
  // the retry overwrites them.  The RNG tick stays consumed, like the reference's already-drawn numbers.
  s->undo_valid = false;
  s->list_valid = false;   // (the carried list is the rolled-back successor's)
  if (s->undo_multipass) return multipass_rollback(s);   // committed in place: the pre-move state is copied back
  s->Xcur = s->undo_X;
  s->vcur ^= 1;
  s->scur ^= 1;
  return 0;
}

int mjhmc_get_tick(mjhmc_sampler* s, uint64_t* tick) {
  if (!s || !tick) return fail(MJHMC_ERR_INVALID, "NULL argument");
  *tick = s->tick;
  return 0;
}

int mjhmc_set_tick(mjhmc_sampler* s, uint64_t tick) {
  if (!s) return fail(MJHMC_ERR_INVALID, "sampler is NULL");
  s->tick = tick;
  s->undo_valid = false;
  return 0;
}

int mjhmc_advance_tick(mjhmc_sampler* s, int64_t n) {
  if (!s || n < 0) return fail(MJHMC_ERR_INVALID, "bad argument");
  s->tick += (uint64_t)n;
  return 0;
}

int mjhmc_reset_flf_cache(mjhmc_sampler* s) {
  if (!s) return fail(MJHMC_ERR_INVALID, "sampler is NULL");
  HIPCHK(hipSetDevice(s->ctx->device));
  HIPCHK(hipMemsetAsync(s->Hflf[s->scur], 0xFF, s->Npad * ssize(s), s->stream));
  s->list_valid = false;
  return drop_spec(s);
}

}  // extern "C"

// per-attempt counters of one mjhmc_iterate call from the device tallies hs[attempt][4]
static void fill_iter_stats(const mjhmc_sampler* s, const std::vector<long long>& hs, int attempts, int done, bool failed,
                            mjhmc_iter_stats* per_iter) {
  if (!per_iter) return;
  for (int i = 0; i < attempts; ++i) {
    mjhmc_iter_stats& st = per_iter[i];
    std::memset(&st, 0, sizeof(st));
    if (s->mode == MJHMC_MODE_MJHMC) {
      st.l = hs[4 * i + 0];
      st.f = hs[4 * i + 1];
      st.r = hs[4 * i + 2];
      st.n_cold = hs[4 * i + 3] & 0xFFFFFFFFLL;   // (the dense kernels: integrated inverse-L trajectories in the high half)
    } else if (s->mode == MJHMC_MODE_CTHMC) {  // clocks FL, F, R (markov_jump_hmc.py:288-290)
      st.fl = hs[4 * i + 0];
      st.f = hs[4 * i + 1];
      st.r = hs[4 * i + 2];
    } else {  // markov_jump_hmc.py:138-148
      st.l = hs[4 * i + 0];
      st.f = hs[4 * i + 1];
      st.r = hs[4 * i + 2];
      st.fl = hs[4 * i + 3];
    }
    st.E_evals = s->N + st.n_cold;  // L on all (+ FLF on the cold ones)
    st.dEdX_evals = (int64_t)s->L * (s->N + st.n_cold);
    st.nonfinite = (failed && i == done) ? 1 : 0;
    st.L_used = s->L;
    st.eps_used = s->eps;
    // inverse-L trajectories integrated: all of the cold ones -- the dense kernels skip the F-movers' (dense_pot.hip)
    st.n_flf_run = (s->mode == MJHMC_MODE_MJHMC && s->en->is_dense()) ? (hs[4 * i + 3] >> 32) : st.n_cold;
  }
}

// Elementwise energies, counter RNG, n_iter >= 2: launches of up to kMaxFuse FUSED iterations.  A launch reads
// the state buffers of one parity and writes the other, so its input survives it: when some particle meets a
// non-finite rate at iteration f (the reference aborts the WHOLE batch there, markov_jump_hmc.py:376-389) the
// launch that contains f is run again for the iterations before f and the call returns n_done = f.
// A big dense batch (every sampler mode) is launched as TWO halves on two streams that run FREELY for the whole
// mjhmc_iterate call.
// Every kernel of the dense path is a persistent grid of one workgroup per CU, so a launch ends with a partial round
// (ProductOfT C3: 3125 tiles = 12.2 rounds of 256, then the inverse-L pass of ~250 tiles another partial one: 14 rounds
// where 13.2 would do) and the machine drains at every launch boundary.  Two independent halves fill each other's
// tails and never drain together.  Results cannot depend on the split (per-particle work, RNG keyed by the global
// particle id; tests/test_gpu_dense_parity.py::test_split_launches_equal_single_launches).
// What the halves must NOT do is meet a non-finite rate while one of them is iterations ahead of the other: the
// roll-back contract of mjhmc_iterate is "the state after `done` iterations", and with two state buffers an iteration
// f+1 of the half that ran ahead has overwritten its own state of iteration f-1.  So a multi-iteration call first
// copies the state aside (one device-to-device copy per call, < 0.2 % of it), and if the flag comes up it puts the copy
// back and runs the call again the single-stream way, which stops at the failing iteration exactly as it always did.
// Measured: two free-running samplers of half the batch each gain 7.8 % (C3) / 5.5 % (C5) over one; halves that met at
// every iteration boundary (the first form of this) kept 1.5 % / 4 %: the dispatcher hands free CUs to the pending
// workgroups of the OLDEST dispatch first, a persistent grid of one workgroup per CU shuts out even a one-block memset
// until a workgroup exits, and a boundary at which both halves wait for each other is a drain again.
// part `which` of a dense batch: particles [start, start + n), their rows of every per-particle array, their stretch of the
// lists, their own three rotating counters (iteration i reads slot i % 3, appends to (i + 1) % 3, clears (i + 2) % 3)
template <class A, typename S>
static A part_args(const A& a, int64_t start, int64_t n, int64_t npad, size_t pitch, int which, int iter, int* counters) {
  A h = a;
  h.X_in = a.X_in + (size_t)start * pitch;
  h.V_in = a.V_in + (size_t)start * pitch;
  h.X_out = a.X_out + (size_t)start * pitch;
  h.V_out = a.V_out + (size_t)start * pitch;
  h.EX_in = a.EX_in + start;
  h.EV_in = a.EV_in + start;
  h.Hflf_in = a.Hflf_in + start;
  h.Hspec_in = a.Hspec_in + start;
  h.Hwork = a.Hwork + start;
  h.EX_out = a.EX_out + start;
  h.EV_out = a.EV_out + start;
  h.Hflf_out = a.Hflf_out + start;
  h.Hspec_out = a.Hspec_out + start;
  h.dwell = a.dwell + start;
  h.dwell_ring = a.dwell_ring + start;
  h.trans = a.trans + start;
  h.cold_list = a.cold_list + start;
  h.next_list = a.next_list + start;
  h.cold_count = counters + 3 * which + iter % 3;
  h.next_count = counters + 3 * which + (iter + 1) % 3;
  h.zero_count = counters + 3 * which + (iter + 2) % 3;
  if constexpr (std::is_same<A, Pot64JumpArgs>::value || std::is_same<A, PotJumpArgs>::value) {
    h.G_in = a.G_in + (size_t)start * pitch;
    h.G_out = a.G_out + (size_t)start * pitch;
  }
  if constexpr (std::is_same<A, Pot64JumpArgs>::value)   // the tile kernel's working rows: one set per concurrent launch
    h.scratch = a.scratch + (size_t)which * pot64_scratch_workgroups() * 2 * 32 * (size_t)pitch;
  h.N = n;
  h.Npad = npad;
  h.first_pid = a.first_pid + start;
  return h;
}

// The parts of a dense batch on their streams (part 0: the sampler's own)
static int ensure_part_streams(mjhmc_sampler* s, int n_parts) {
  if (!s->ev_fork) {
    HIPCHK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
  }
  while ((int)s->part_streams.size() < n_parts - 1) {
    hipStream_t q;
    hipEvent_t e;
    HIPCHK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
    s->part_streams.push_back(q);
    HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    s->part_events.push_back(e);
  }
  return 0;
}

// the state a multi-iteration split call starts from, copied aside / put back (the layout of mjhmc_checkpoint, own buffers)
static int split_state_copy(mjhmc_sampler* s, bool restore) {
  const size_t mb = mat_bytes(s), vb = (size_t)s->Npad * ssize(s), db = (size_t)s->Npad * sizeof(double);
  const int nck = s->en->is_pot() ? 7 : 6;
  const size_t sizes[7] = {mb, mb, vb, vb, vb, db, mb};
  void* live[7] = {s->Xcur, s->Vbuf[s->vcur], s->EX[s->scur], s->EV[s->scur], s->Hflf[s->scur], s->dwell, s->Gbuf[s->vcur]};
  for (int i = 0; i < nck; ++i) {
    if (!s->ick[i]) HIPCHK(hipMalloc(&s->ick[i], sizes[i]));
    if (restore) HIPCHK(hipMemcpyAsync(live[i], s->ick[i], sizes[i], hipMemcpyDeviceToDevice, s->stream));
    else HIPCHK(hipMemcpyAsync(s->ick[i], live[i], sizes[i], hipMemcpyDeviceToDevice, s->stream));
  }
  // (Hspec follows scur and an attempt only writes the other parity, so the call's starting values survive a failed
  // attempt like H_flf's do -- but the halves may have run several iterations: both parities are overwritten)
  if (restore) TRY(drop_spec(s));
  return 0;
}

// failure flag + cold-list counters + tallies of a call: ONE copy of the head of the call block into pinned host memory
// (a pageable destination makes a small copy a synchronous staged transfer: ~20 us) and ONE stream sync.  Until round 6
// the call was three fills, two asynchronous copies and -- for the compacted passes -- a synchronous third.
static int read_back_call(mjhmc_sampler* s, size_t stats_rows, Control* hc, long long* hs, int* counts = nullptr, int n_counts = 0) {
  const size_t head = kCtlBytes + counts_bytes(s->stats_cap);
  const size_t need = head + stats_rows * 4 * sizeof(long long);
  if (s->h_pin_cap < need) {
    if (s->h_pin) HIPCHK(hipHostFree(s->h_pin));
    s->h_pin = nullptr;
    s->h_pin_cap = 0;
    HIPCHK(hipHostMalloc((void**)&s->h_pin, need * 2, hipHostMallocDefault));
    s->h_pin_cap = need * 2;
  }
  HIPCHK(hipMemcpyAsync(s->h_pin, s->call_block, need, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  std::memcpy(hc, s->h_pin, sizeof(Control));
  if (counts) std::memcpy(counts, s->h_pin + kCtlBytes, (size_t)n_counts * sizeof(int));
  std::memcpy(hs, s->h_pin + head, stats_rows * 4 * sizeof(long long));
  return 0;
}

template <typename T>
static int iterate_fused_t(mjhmc_sampler* s, int n_iter, int ring_slot0, mjhmc_iter_stats* per_iter, int* n_done) {
  const size_t mb = mat_bytes(s);
  const int need = n_iter + kMaxFuse;  // + scratch tallies for the recovery launch
  TRY(zero_call(s, need));

  void* xin = s->Xcur;
  if (ring_slot0 >= 0) {  // the live state sits in a slot about to be overwritten: move it out first
    const char* lo = (const char*)s->ring + (size_t)ring_slot0 * mb;
    const char* hi = lo + (size_t)n_iter * mb;
    if ((const char*)xin >= lo && (const char*)xin < hi) {
      void* spare = s->Xbuf[0];
      HIPCHK(hipMemcpyAsync(spare, xin, mb, hipMemcpyDeviceToDevice, s->stream));
      xin = s->Xcur = spare;
    }
  }
  struct Launch {
    int i0, K, vi, si;
    void* xin;
    void* xout;
  };
  int64_t split_at = 0;  // particles per part
  int n_parts = 3;       // C2: 2 / 3 / 4 / 8 parts measured 0.0803 / 0.0779 / 0.0803 / 0.0782 ms per iteration, unsplit 0.0855
  bool allow_split_now = true;  // (the recovery launch runs on one stream)
  if (ring_slot0 < 0 && !test_env("MJHMC_NO_SPLIT")) {
    if (const char* np = test_env("MJHMC_SPLIT_PARTS")) n_parts = std::max(2, std::min(8, std::atoi(np)));
    const int64_t nslots = s->Npad >> (6 - s->sh.logG);
    if (nslots >= 8 * 4096 && !fused_rows(s)) split_at = (s->Npad / n_parts) / 256 * 256;  // whole workgroups' worth of slots in every part
    if (split_at <= 0 || split_at * (n_parts - 1) >= s->N) split_at = 0;
    if (split_at) TRY(ensure_part_streams(s, n_parts));
  }
  auto launch = [&](const Launch& l, long long* stats) -> int {
    JumpArgs<T> a;
    a.X_in = (const T*)l.xin;
    a.V_in = (const T*)s->Vbuf[l.vi];
    a.X_out = (T*)l.xout;
    a.V_out = (T*)s->Vbuf[l.vi ^ 1];
    a.EX_in = (const T*)s->EX[l.si];
    a.EV_in = (const T*)s->EV[l.si];
    a.EX_out = (T*)s->EX[l.si ^ 1];
    a.EV_out = (T*)s->EV[l.si ^ 1];
    a.Hflf_in = (const T*)s->Hflf[l.si];
    a.Hflf_out = (T*)s->Hflf[l.si ^ 1];
    a.dwell = s->dwell;
    a.trans = s->trans;
    a.noise = nullptr;
    a.rexp = nullptr;
    a.runif = nullptr;
    a.mode = s->mode;
    a.ctl = s->ctl;
    a.stats = (unsigned long long*)stats;
    a.N = s->N;
    a.Npad = s->Npad;
    a.first_pid = s->first_pid;
    a.D = s->D;
    a.pitch = s->sh.pitch;
    a.CH = s->sh.CH;
    a.logG = s->sh.logG;
    a.L = s->L;
    a.iter = l.i0;
    a.defer_r = 0;
    a.n_fuse = l.K;
    a.ab = ab_flags();
    if (ring_slot0 >= 0) {
      a.xiter = (T*)((char*)s->ring + (size_t)(ring_slot0 + l.i0) * mb);
      a.xiter_stride = mb / sizeof(T);
      a.dwell_ring = s->dwell_ring + (size_t)(ring_slot0 + l.i0) * s->Npad;
    } else {
      a.xiter = nullptr;
      a.xiter_stride = 0;
      a.dwell_ring = s->dwell_scratch;
    }
    a.eps = (T)s->eps;
    a.chalf = (T)(-s->eps / 2.);
    a.r_keep = (T)std::sqrt(1. - s->beta);
    a.r_mix = (T)std::sqrt(s->beta);
    a.p_r = s->p_r;
    a.p_flip = s->p_flip;
    const uint64_t tick = s->tick + (uint64_t)l.i0;
    a.key = RngKey{(uint32_t)(s->seed & 0xFFFFFFFFu), (uint32_t)(s->seed >> 32), (uint32_t)(tick & 0xFFFFFFFFu),
                   (uint32_t)(tick >> 32)};
    if (split_at > 0 && allow_split_now) {
      // several parts on as many streams: a launch is a persistent grid of one slot (here: up to 64 iterations of a
      // particle) per wave and pass, so it ends with a partial pass (C2: 100 000 slots on 4096 resident waves = 24.4
      // passes, 25 paid), and its workgroups all sit in the same phase of their slots at the same time; the parts'
      // workgroups start on the CUs the earlier parts leave and run out of step with them.  The parts meet at the end of
      // every fused launch (once per 64 iterations): the recovery protocol below relies on the inputs of the failing
      // launch being intact, i.e. on no part having started the next launch.
      HIPCHK(hipEventRecord(s->ev_fork, s->stream));
      for (int k = 0; k < n_parts; ++k) {
        const int64_t start = (int64_t)k * split_at, stop = (k + 1 == n_parts) ? a.N : start + split_at;
        JumpArgs<T> h = a;
        const size_t po = (size_t)start * a.pitch;
        h.X_in = a.X_in + po;
        h.V_in = a.V_in + po;
        h.X_out = a.X_out + po;
        h.V_out = a.V_out + po;
        h.EX_in = a.EX_in + start;
        h.EV_in = a.EV_in + start;
        h.EX_out = a.EX_out + start;
        h.EV_out = a.EV_out + start;
        h.Hflf_in = a.Hflf_in + start;
        h.Hflf_out = a.Hflf_out + start;
        h.dwell = a.dwell + start;
        h.dwell_ring = a.dwell_ring + start;
        h.trans = a.trans + start;
        h.first_pid = a.first_pid + start;
        h.N = stop - start;
        h.Npad = (k + 1 == n_parts) ? a.Npad - start : split_at;
        hipStream_t q = k == 0 ? s->stream : s->part_streams[(size_t)k - 1];
        if (k > 0) HIPCHK(hipStreamWaitEvent(q, s->ev_fork, 0));
        TRY(dispatch_jump<T>(s->en->ep.kind, h, s->en->ep, s->sh.E, q));
        if (k > 0) {
          HIPCHK(hipEventRecord(s->part_events[(size_t)k - 1], q));
          HIPCHK(hipStreamWaitEvent(s->stream, s->part_events[(size_t)k - 1], 0));
        }
      }
    } else {
      TRY(dispatch_jump<T>(s->en->ep.kind, a, s->en->ep, s->sh.E, s->stream));
    }
    HIPCHK(hipGetLastError());
    return 0;
  };
  auto out_of = [&](int i0, int K, void* in) -> void* {
    if (ring_slot0 >= 0) return (char*)s->ring + (size_t)(ring_slot0 + i0 + K - 1) * mb;
    return (in != s->Xbuf[0]) ? s->Xbuf[0] : s->Xbuf[1];
  };

#ifdef MJHMC_TEST_HOOKS
  if (const char* poison = test_env("MJHMC_DEBUG_POISON")) {  // test hook, see iterate_t (fused launches: iteration 0 only)
    int pit = -1;
    long long pp = -1;
    if (std::sscanf(poison, "%d:%lld", &pit, &pp) == 2 && pit == 0 && pp >= 0 && pp < s->N) {
      static const double nan64 = __builtin_nan("");
      static const float nan32 = __builtin_nanf("");
      HIPCHK(hipMemcpyAsync((char*)s->EV[s->scur] + (size_t)pp * ssize(s), sizeof(T) == 8 ? (const void*)&nan64 : (const void*)&nan32,
                            ssize(s), hipMemcpyHostToDevice, s->stream));
    }
  }
#endif  // MJHMC_TEST_HOOKS
  std::vector<Launch> launches;
  if (s->timing_on) HIPCHK(hipEventRecord(s->ev_total[0], s->stream));
  for (int i0 = 0; i0 < n_iter; i0 += kMaxFuse) {
    const int j = (int)launches.size();
    Launch l;
    l.i0 = i0;
    l.K = std::min(kMaxFuse, n_iter - i0);
    l.vi = (s->vcur + j) & 1;
    l.si = (s->scur + j) & 1;
    l.xin = xin;
    l.xout = out_of(i0, l.K, xin);
    TRY(launch(l, s->stats + 4 * (size_t)i0));
    launches.push_back(l);
    xin = l.xout;
  }
  if (s->timing_on) HIPCHK(hipEventRecord(s->ev_total[1], s->stream));
  Control hc;
  std::vector<long long> hs((size_t)n_iter * 4);
  TRY(read_back_call(s, (size_t)n_iter, &hc, hs.data()));

  int done = n_iter, attempts = n_iter, committed = (int)launches.size();
  void* xlive = launches.back().xout;
  if (hc.failed) {
    done = 0x7fffffff - hc.inv_iter;
    attempts = done + 1;
    committed = done / kMaxFuse;
    Launch l = launches[(size_t)committed];
    xlive = l.xin;
    l.K = done - l.i0;
    if (l.K > 0) {  // redo the iterations of that launch that precede the failed one (same ticks, same results)
      l.xout = out_of(l.i0, l.K, l.xin);
      long long* redo_stats = s->stats + 4 * (size_t)n_iter;
      HIPCHK(hipMemsetAsync(s->ctl, 0, sizeof(Control), s->stream));
      HIPCHK(hipMemsetAsync(redo_stats, 0, (size_t)kMaxFuse * 4 * sizeof(long long), s->stream));
      allow_split_now = false;
      TRY(launch(l, redo_stats));
      Control rc;
      std::vector<long long> rs((size_t)l.K * 4);
      HIPCHK(hipMemcpyAsync(&rc, s->ctl, sizeof(Control), hipMemcpyDeviceToHost, s->stream));
      HIPCHK(hipMemcpyAsync(rs.data(), redo_stats, rs.size() * sizeof(long long), hipMemcpyDeviceToHost, s->stream));
      HIPCHK(hipStreamSynchronize(s->stream));
      if (rc.failed)  // cannot happen: the same ticks on the same inputs gave finite rates for these iterations before
        return fail(MJHMC_ERR_HIP, "the recovery launch of a fused batch reported a non-finite rate of its own");
      // the tallies of the committed iterations of that launch come from the recovery run: in the failed launch a
      // workgroup that started after the failure may have skipped them
      for (int i = 0; i < l.K; ++i)
        for (int c = 0; c < 4; ++c) hs[(size_t)(l.i0 + i) * 4 + c] = rs[(size_t)i * 4 + c];
      xlive = l.xout;
      ++committed;
    }
  }
  fill_iter_stats(s, hs, attempts, done, hc.failed != 0, per_iter);
  s->undo_valid = false;
  s->Xcur = xlive;
  s->vcur = (s->vcur + committed) & 1;
  s->scur = (s->scur + committed) & 1;
  s->tick += (uint64_t)attempts;
  if (n_done) *n_done = done;
  s->last_jump_launches = attempts;
  s->timing_pending = s->timing_on;
  return 0;
}

// batches of the non-Gaussian elementwise energies below this many particles run fused (measured on the funnel with a group
// of lanes per particle, tools/sweep_shard_c4.py `fused_groups` / `groups`: at 125 000 particles 0.047 ms per iteration fused,
// 0.051 as trajectory + jump-process launches; at 250 000 0.094 against 0.082).  The funnels themselves no longer come here:
// their float64 rows of 9 ... 32 dims fuse in row form at every size (fused_rows).
constexpr int64_t kFuseBelow = 160000;

template <typename T>
static int iterate_t(mjhmc_sampler* s, int n_iter, const double* replay_normal, const double* replay_exp,
                     const double* replay_unif, int ring_slot0, mjhmc_iter_stats* per_iter, int* n_done,
                     bool allow_split = true) {
  // Fused launches: always for the Gaussian forces (one iteration is HBM-bound), for the other elementwise energies
  // while the batch is small (launch-/latency-bound) -- big batches of those take the compacted passes below instead.
  const bool carried = s->list_valid && !test_env("MJHMC_NO_LIST_CARRY");   // (every path consumes it; only the compacted passes renew it)
  s->list_valid = false;
  const bool gaussian = s->en->ep.kind == MJHMC_E_ISO_GAUSS || s->en->ep.kind == MJHMC_E_DIAG_GAUSS;
  int64_t fuse_below = kFuseBelow;
  if (const char* fb = test_env("MJHMC_FUSE_BELOW")) fuse_below = std::atoll(fb);
  // (the funnels in row form -- a lane per particle, mjhmc_fused_rows_kernel -- are bound by the vector pipe at any batch size)
  const bool rows = fused_rows(s) && !test_env("MJHMC_FUSE_BELOW");
  const bool fusable = !s->en->is_dense() && !s->en->is_user() && (gaussian || rows || s->N < fuse_below || s->D <= 4);
  // (Measured in round 6 and dropped: a call of ONE iteration through the fused row kernel too -- the state read and written
  // once with the jump process inside the launch, no second launch, no list scan: 0.346 ms per call against the 0.270 of
  // trajectory launch + jump-process launch at C4's size, equal at N/8.  One wave per SIMD overlaps nothing of a tile's
  // loads and stores with its compute, and a one-iteration launch is nothing but tiles' first and last iterations.)
  if (n_iter >= 2 && fusable && !replay_normal && !replay_exp && !replay_unif && !test_env("MJHMC_NO_FUSE"))
    return iterate_fused_t<T>(s, n_iter, ring_slot0, per_iter, n_done);
  const size_t mb = mat_bytes(s);
  TRY(ensure_call_block(s, n_iter));
  if (replay_normal && !s->noise) {
    HIPCHK(hipMalloc(&s->noise, mb));
    HIPCHK(hipMemsetAsync(s->noise, 0, mb, s->stream));
  }
  if (replay_exp && !s->rexp) HIPCHK(hipMalloc((void**)&s->rexp, 3 * s->N * sizeof(double)));
  if (replay_unif && !s->runif) HIPCHK(hipMalloc((void**)&s->runif, (2 * s->N + 1) * sizeof(double)));
  TRY(zero_call(s, n_iter));   // (failure flag, the cold lists' counters, the tallies: one fill)

  // Several particles per wave and a big batch: the inverse-L trajectory of the cold-cache particles runs in its
  // the list's own workgroups of mjhmc_traj_kernel instead of in every wave of the jump kernel that holds a cold particle.
  const bool compact = s->mode == MJHMC_MODE_MJHMC && !s->en->is_dense() && !s->en->is_user() && !replay_normal && !replay_exp &&
                       s->sh.logG < 6 && s->N >= 16384 && !test_env("MJHMC_NO_COMPACT");   // (reached from kFuseBelow particles up, or MJHMC_NO_FUSE)
  if (compact) {
    if (!s->flf_list) {
      HIPCHK(hipMalloc((void**)&s->flf_list, (size_t)2 * s->Npad * sizeof(int)));   // two lists: iterations alternate
      HIPCHK(hipMalloc(&s->Hpre, (size_t)2 * s->Npad * ssize(s)));
    }
    // (flf_counts: a counter per iteration of the call + the one the last iteration's movers go to -- stats_cap + 1 of them
    // in the call block, zeroed with it)
  }

  // dense batches: free-running parts on their own streams (part_args; the story is above iterate_fused_t)
  int n_parts = 1;
  int64_t part_len = 0;   // particles per part (the last part takes the rest)
  // (The elementwise compacted passes -- C4 -- were tried as 2 / 3 / 4 free-running parts the same way: 0.2727 ms per
  // iteration unsplit, 0.2729 / 0.2744 / 0.310 split: their kernels leave room for each other already.)
  if (s->en->is_dense()) {
    const int cus = std::max(1, s->ctx->prop.multiProcessorCount);  // of THIS sampler's device
    const int ppt = s->en->is_sic() ? sic_particles_per_tile(s->en->sic_P) : 32;
    const int64_t unit = 64 * (int64_t)ppt;  // whole tiles and whole 64-particle row groups in every part
    const int64_t ntiles = (s->N + ppt - 1) / ppt;
    // A launch is a persistent grid of one workgroup per CU whose items (forward tiles + the inverse-L tiles of the R-movers)
    // take one trajectory time each: alone it ends in a partial round, and at a few hundred tiles that round is most of
    // the launch (C3 at 12 500 particles: 410 items = 1.6 rounds run as 2).  Two parts that never wait for each other fill
    // each other's partial rounds; below ~3/4 of a round of tiles there is nothing to fill.
    // (recording into ring slots changes nothing: iteration i reads slot i - 1 and writes slot i, part by part)
    int want = (allow_split && !replay_normal && !replay_exp && !replay_unif && !test_env("MJHMC_NO_SPLIT") &&
                4 * ntiles >= 3 * (int64_t)cus) ? 2 : 1;
    if (want > 1)
      if (const char* np = test_env("MJHMC_SPLIT_PARTS")) want = std::max(1, std::min(kMaxDenseParts, std::atoi(np)));
    if (want > 1) {
      part_len = (s->Npad / want) / unit * unit;
      if (part_len > 0 && part_len * (want - 1) < s->N) n_parts = want;
    }
    if (n_parts > 1) {
      TRY(ensure_part_streams(s, n_parts));
      if (n_iter > 1) TRY(split_state_copy(s, false));
    }
  }
  // test build, MJHMC_NO_FSPEC=1: nothing is handed on from an F move to the next iteration -- every cold particle's
  // inverse-L proposal is integrated, as the reference does (the A side of test_f_mover_shortcut_is_bit_identical)
  void* spec_out_override = nullptr;
  if (s->en->is_dense() && s->mode == MJHMC_MODE_MJHMC && test_env("MJHMC_NO_FSPEC")) {
    if (!s->Hspec_dump) HIPCHK(hipMalloc(&s->Hspec_dump, (size_t)s->Npad * ssize(s)));
    for (int i = 0; i < 2; ++i) HIPCHK(hipMemsetAsync(s->Hspec[i], 0xFF, (size_t)s->Npad * ssize(s), s->stream));
    spec_out_override = s->Hspec_dump;
  }
  auto part_stream = [&](int k) { return k == 0 ? s->stream : s->part_streams[(size_t)k - 1]; };
  auto part_start = [&](int k) { return (int64_t)k * part_len; };
  auto part_count = [&](int k) { return k + 1 == n_parts ? s->N - part_start(k) : part_len; };
  auto part_npad = [&](int k) { return k + 1 == n_parts ? s->Npad - part_start(k) : part_len; };
  int* const dense_counters = s->cold_list ? s->cold_list + 2 * s->Npad : nullptr;

  std::vector<void*> xout(n_iter);
  void* xin = s->Xcur;
  if (s->timing_on) HIPCHK(hipEventRecord(s->ev_total[0], s->stream));
  for (int i = 0; i < n_iter; ++i) {
    void* xo;
    double* dring = s->dwell_scratch;
    if (ring_slot0 >= 0) {
      xo = (char*)s->ring + (size_t)(ring_slot0 + i) * mb;
      dring = s->dwell_ring + (size_t)(ring_slot0 + i) * s->Npad;
      if (xo == xin) {  // the live state sits in the slot about to be overwritten: move it out first
        void* spare = s->Xbuf[0];
        HIPCHK(hipMemcpyAsync(spare, xin, mb, hipMemcpyDeviceToDevice, s->stream));
        xin = spare;
        if (i == 0) s->Xcur = spare;
      }
    } else {
      xo = (xin != s->Xbuf[0]) ? s->Xbuf[0] : s->Xbuf[1];
    }
    xout[i] = xo;
    const int vi = (s->vcur + i) & 1, si = (s->scur + i) & 1;
    if (replay_normal) TRY(upload_matrix(s, replay_normal + (size_t)i * s->D * s->N, s->noise));
    if (replay_exp)
      HIPCHK(hipMemcpyAsync(s->rexp, replay_exp + (size_t)i * 3 * s->N, 3 * s->N * sizeof(double),
                            hipMemcpyHostToDevice, s->stream));
    if (replay_unif)
      HIPCHK(hipMemcpyAsync(s->runif, replay_unif + (size_t)i * (2 * s->N + 1), (2 * s->N + 1) * sizeof(double),
                            hipMemcpyHostToDevice, s->stream));
#ifdef MJHMC_TEST_HOOKS
    if (const char* poison = test_env("MJHMC_DEBUG_POISON")) {
      // test hook "iteration:particle": that particle's kinetic energy reads NaN in that iteration of the call -> its rates
      // are not finite -> the whole-batch abort of markov_jump_hmc.py:376-389, at a chosen point of a multi-iteration call
      int pit = -1;
      long long pp = -1;
      if (std::sscanf(poison, "%d:%lld", &pit, &pp) == 2 && pit == i && pp >= 0 && pp < s->N) {
        static const double nan64 = __builtin_nan("");
        static const float nan32 = __builtin_nanf("");
        const int pk = n_parts > 1 ? (int)std::min<int64_t>(pp / part_len, n_parts - 1) : 0;
        hipStream_t pst = part_stream(pk);
        if (pk > 0 && i == 0) HIPCHK(hipStreamSynchronize(s->stream));  // (the fork below comes later)
        HIPCHK(hipMemcpyAsync((char*)s->EV[si] + (size_t)pp * ssize(s), sizeof(T) == 8 ? (const void*)&nan64 : (const void*)&nan32,
                              ssize(s), hipMemcpyHostToDevice, pst));
      }
    }
#endif  // MJHMC_TEST_HOOKS
    if (n_parts > 1 && i == 0) {  // the other parts' streams start from everything the first has been given so far
      HIPCHK(hipEventRecord(s->ev_fork, s->stream));
      for (int k = 1; k < n_parts; ++k) HIPCHK(hipStreamWaitEvent(part_stream(k), s->ev_fork, 0));
    }
    JumpArgs<T> a;
    a.X_in = (const T*)xin;
    a.V_in = (const T*)s->Vbuf[vi];
    a.X_out = (T*)xo;
    a.V_out = (T*)s->Vbuf[vi ^ 1];
    a.EX_in = (const T*)s->EX[si];
    a.EV_in = (const T*)s->EV[si];
    a.EX_out = (T*)s->EX[si ^ 1];
    a.EV_out = (T*)s->EV[si ^ 1];
    a.Hflf_in = (const T*)s->Hflf[si];
    a.Hflf_out = (T*)s->Hflf[si ^ 1];
    a.dwell = s->dwell;
    a.dwell_ring = dring;
    a.trans = s->trans;
    a.noise = replay_normal ? (const T*)s->noise : nullptr;
    a.rexp = replay_exp ? s->rexp : nullptr;
    a.runif = replay_unif ? s->runif : nullptr;
    a.mode = s->mode;
    a.ctl = s->ctl;
    a.stats = (unsigned long long*)(s->stats + 4 * i);
    a.N = s->N;
    a.Npad = s->Npad;
    a.first_pid = s->first_pid;
    a.D = s->D;
    a.pitch = s->sh.pitch;
    a.CH = s->sh.CH;
    a.logG = s->sh.logG;
    a.L = s->L;
    a.iter = i;
    a.n_fuse = 0;
    a.defer_r = 0;
    a.ab = ab_flags();
    a.xiter = nullptr;
    a.xiter_stride = 0;
    a.eps = (T)s->eps;
    a.chalf = (T)(-s->eps / 2.);
    a.r_keep = (T)std::sqrt(1. - s->beta);
    a.r_mix = (T)std::sqrt(s->beta);
    a.p_r = s->p_r;
    a.p_flip = s->p_flip;
    const uint64_t tick = s->tick + (uint64_t)i;
    a.key = RngKey{(uint32_t)(s->seed & 0xFFFFFFFFu), (uint32_t)(s->seed >> 32), (uint32_t)(tick & 0xFFFFFFFFu),
                   (uint32_t)(tick >> 32)};
    if (s->en->is_pot()) {
      if constexpr (sizeof(T) == 4) {
        PotJumpArgs pa;
        pa.X_in = a.X_in;
        pa.V_in = a.V_in;
        pa.G_in = (const float*)s->Gbuf[vi];
        pa.X_out = a.X_out;
        pa.V_out = a.V_out;
        pa.G_out = (float*)s->Gbuf[vi ^ 1];
        pa.EX_in = a.EX_in;
        pa.EV_in = a.EV_in;
        pa.Hflf_in = a.Hflf_in;
        pa.Hwork = s->Hwork;
        pa.cold_list = s->cold_list + (size_t)(i & 1) * s->Npad;          // iteration i reads list i & 1 ...
        pa.next_list = s->cold_list + (size_t)((i + 1) & 1) * s->Npad;    // ... and writes the next iteration's
        pa.Hspec_in = (const float*)s->Hspec[si];
        pa.Hspec_out = (float*)(spec_out_override ? spec_out_override : s->Hspec[si ^ 1]);
        pa.rescan = spec_out_override ? 1 : 0;
        pa.EX_out = a.EX_out;
        pa.EV_out = a.EV_out;
        pa.Hflf_out = a.Hflf_out;
        pa.dwell = a.dwell;
        pa.dwell_ring = a.dwell_ring;
        pa.trans = a.trans;
        pa.noise = a.noise;
        pa.rexp = a.rexp;
        pa.runif = a.runif;
        pa.mode = a.mode;
        pa.p_flip = a.p_flip;
        pa.ctl = a.ctl;
        pa.stats = a.stats;
        pa.N = a.N;
        pa.Npad = a.Npad;
        pa.ntiles = a.Npad / 32;
        pa.first_pid = a.first_pid;
        pa.D = a.D;
        pa.L = a.L;
        pa.iter = a.iter;
        pa.eps = a.eps;
        pa.chalf = a.chalf;
        pa.r_keep = a.r_keep;
        pa.r_mix = a.r_mix;
        pa.p_r = a.p_r;
        pa.key = a.key;
        for (int k = 0; k < n_parts; ++k) {
          PotJumpArgs h = part_args<PotJumpArgs, float>(pa, n_parts > 1 ? part_start(k) : 0, n_parts > 1 ? part_count(k) : s->N,
                                                        n_parts > 1 ? part_npad(k) : s->Npad, (size_t)s->sh.pitch, k, i, dense_counters);
          h.ntiles = h.Npad / 32;
          pot_launch_jump(h, s->en->pot_model(), part_stream(k));
        }
      } else {
        // the reference's arithmetic: float64 state rows streamed through the tile kernel's epilogue (dense_pot64.hip)
        Pot64JumpArgs pa;
        pa.X_in = (const double*)a.X_in;
        pa.V_in = (const double*)a.V_in;
        pa.G_in = (const double*)s->Gbuf[vi];
        pa.X_out = (double*)a.X_out;
        pa.V_out = (double*)a.V_out;
        pa.G_out = (double*)s->Gbuf[vi ^ 1];
        pa.EX_in = (const double*)a.EX_in;
        pa.EV_in = (const double*)a.EV_in;
        pa.Hflf_in = (const double*)a.Hflf_in;
        pa.Hwork = (double*)s->Hwork;
        pa.cold_list = s->cold_list + (size_t)(i & 1) * s->Npad;          // iteration i reads list i & 1 ...
        pa.next_list = s->cold_list + (size_t)((i + 1) & 1) * s->Npad;    // ... and writes the next iteration's
        pa.Hspec_in = (const double*)s->Hspec[si];
        pa.Hspec_out = (double*)(spec_out_override ? spec_out_override : s->Hspec[si ^ 1]);
        pa.rescan = spec_out_override ? 1 : 0;
        pa.EX_out = (double*)a.EX_out;
        pa.EV_out = (double*)a.EV_out;
        pa.Hflf_out = (double*)a.Hflf_out;
        pa.dwell = a.dwell;
        pa.dwell_ring = a.dwell_ring;
        pa.trans = a.trans;
        pa.noise = (const double*)a.noise;
        pa.rexp = a.rexp;
        pa.runif = a.runif;
        pa.scratch = s->pot64_scratch;
        pa.mode = a.mode;
        pa.p_flip = a.p_flip;
        pa.ctl = a.ctl;
        pa.stats = a.stats;
        pa.N = a.N;
        pa.Npad = a.Npad;
        pa.ntiles = a.Npad / 32;
        pa.first_pid = a.first_pid;
        pa.D = a.D;
        pa.L = a.L;
        pa.iter = a.iter;
        pa.eps = s->eps;
        pa.chalf = -s->eps / 2.;
        pa.r_keep = std::sqrt(1. - s->beta);
        pa.r_mix = std::sqrt(s->beta);
        pa.p_r = a.p_r;
        pa.key = a.key;
        for (int k = 0; k < n_parts; ++k) {
          Pot64JumpArgs h = part_args<Pot64JumpArgs, double>(pa, n_parts > 1 ? part_start(k) : 0, n_parts > 1 ? part_count(k) : s->N,
                                                             n_parts > 1 ? part_npad(k) : s->Npad, (size_t)s->sh.pitch, k, i, dense_counters);
          h.ntiles = h.Npad / 32;
          pot64_launch_jump(h, s->en->pot_model(), part_stream(k));
        }
      }
    } else if (s->en->is_sic()) {
      if constexpr (sizeof(T) == 4) {
        auto launch_sic = [&](auto tag) {
          using ST = decltype(tag);
          SicJumpArgsT<ST> sa;
          sa.X_in = (const ST*)xin;
          sa.V_in = (const ST*)s->Vbuf[vi];
          sa.X_out = (ST*)xo;
          sa.V_out = (ST*)s->Vbuf[vi ^ 1];
          sa.EX_in = a.EX_in;
          sa.EV_in = a.EV_in;
          sa.Hflf_in = a.Hflf_in;
          sa.Hwork = s->Hwork;
          sa.cold_list = s->cold_list + (size_t)(i & 1) * s->Npad;
          sa.next_list = s->cold_list + (size_t)((i + 1) & 1) * s->Npad;
          sa.Hspec_in = (const float*)s->Hspec[si];
          sa.Hspec_out = (float*)(spec_out_override ? spec_out_override : s->Hspec[si ^ 1]);
          sa.rescan = spec_out_override ? 1 : 0;
          sa.EX_out = a.EX_out;
          sa.EV_out = a.EV_out;
          sa.Hflf_out = a.Hflf_out;
          sa.dwell = a.dwell;
          sa.dwell_ring = a.dwell_ring;
          sa.trans = a.trans;
          sa.noise = (const ST*)a.noise;
          sa.rexp = a.rexp;
          sa.runif = a.runif;
          sa.mode = a.mode;
          sa.p_flip = a.p_flip;
          sa.ctl = a.ctl;
          sa.stats = a.stats;
          sa.N = a.N;
          sa.Npad = a.Npad;
          const int ppt = sic_particles_per_tile(s->en->sic_P);
          sa.ntiles = (a.N + ppt - 1) / ppt;
          sa.first_pid = a.first_pid;
          sa.L = a.L;
          sa.iter = a.iter;
          sa.eps = a.eps;
          sa.chalf = a.chalf;
          sa.r_keep = a.r_keep;
          sa.r_mix = a.r_mix;
          sa.p_r = a.p_r;
          sa.key = a.key;
          for (int k = 0; k < n_parts; ++k) {
            SicJumpArgsT<ST> h = part_args<SicJumpArgsT<ST>, ST>(sa, n_parts > 1 ? part_start(k) : 0, n_parts > 1 ? part_count(k) : s->N,
                                                                 n_parts > 1 ? part_npad(k) : s->Npad, (size_t)s->sh.pitch, k, i,
                                                                 dense_counters);
            h.ntiles = (h.N + ppt - 1) / ppt;
            sic_launch_jump(h, s->en->sic_model(), part_stream(k));
          }
        };
        // the state's type: bfloat16 (BASELINE.json configs[4]) or float32 (the reference's TensorFlow float32)
        if (s->dtype == MJHMC_BF16) launch_sic(__bf16{});
        else launch_sic(float{});
      }
    } else {
      if (compact) {
        // elementwise.hpp (mjhmc_step_kernel): the iteration's trajectories -- the L proposals and, beside them, the
        // inverse-L proposals of the listed cold caches --, then its jump process with a lane per particle, which finishes
        // the movers and hands them on as the next iteration's list (only the first iteration of a call scans the cache).
        // (Measured and dropped: the batch as two halves on two free-running streams, 0.2551 ms against 0.2623 for C4 on the
        // device but more launches than the host issues in that time; and the halves staggered by half an iteration inside
        // ONE launch sequence -- every launch the trajectories of one half beside the jump process of the other --,
        // 0.2586 ms: the trajectories keep the vector pipe half busy themselves, there is no idle unit to hide the jump
        // process under.  DESIGN.md section 8c.)
        TrajArgs<T> ta;
        JumpDecideArgs<T> da;
        ta.X_in = a.X_in;
        ta.V_in = a.V_in;
        ta.X_out = a.X_out;
        ta.V_out = a.V_out;
        ta.EX_out = a.EX_out;
        ta.EV_out = a.EV_out;
        ta.Hwork = (T*)s->Hpre;
        // the two lists alternate; a call that follows a committed call of this path starts from the list that call's last
        // jump process left (the host knows its length) instead of scanning H_flf
        const int par0 = carried ? s->list_par : 0;
        ta.list = s->flf_list + (size_t)((par0 + i) & 1) * s->Npad;
        ta.count = (carried && i == 0) ? nullptr : s->flf_counts + i;
        ta.n_listed = s->list_count;
        ta.ctl = s->ctl;
        ta.N = a.N;
        ta.D = a.D;
        ta.pitch = a.pitch;
        ta.CH = a.CH;
        ta.logG = a.logG;
        ta.L = a.L;
        const int64_t ppb = 256 >> a.logG;   // the list's walkers: sized for a typical list (a few per cent of the batch)
        ta.inv_blocks = (int)std::max<int64_t>(1, std::min<int64_t>((a.N + ppb - 1) / ppb / 8, 2048));
        ta.rows = test_env("MJHMC_NO_ROWS") ? 0 : 1;
        ta.eps = a.eps;
        ta.chalf = a.chalf;
        da.X_in = a.X_in;
        da.V_in = a.V_in;
        da.X_out = a.X_out;
        da.V_out = a.V_out;
        da.EX_in = a.EX_in;
        da.EV_in = a.EV_in;
        da.EX_out = a.EX_out;
        da.EV_out = a.EV_out;
        da.Hflf_in = a.Hflf_in;
        da.Hwork = ta.Hwork;
        da.Hflf_out = a.Hflf_out;
        da.dwell = a.dwell;
        da.dwell_ring = a.dwell_ring;
        da.trans = a.trans;
        da.next_list = s->flf_list + (size_t)((par0 + i + 1) & 1) * s->Npad;
        da.next_count = s->flf_counts + i + 1;
        da.ctl = s->ctl;
        da.stats = a.stats;
        da.N = a.N;
        da.first_pid = a.first_pid;
        da.D = a.D;
        da.pitch = a.pitch;
        da.CH = a.CH;
        da.logG = a.logG;
        da.iter = a.iter;
        da.r_keep = a.r_keep;
        da.r_mix = a.r_mix;
        da.p_r = a.p_r;
        da.key = a.key;
        if (i == 0 && !carried) {
          const dim3 list_grid((unsigned)((s->N + kColdChunk - 1) / kColdChunk));
          hipLaunchKernelGGL(compact_list_kernel<ColdCache<T>>, list_grid, dim3(1024), 0, s->stream,
                             ColdCache<T>{a.Hflf_in, (T*)s->Hpre + s->Npad}, s->N, s->ctl, s->flf_list, s->flf_counts);
        }
        TRY(dispatch_step<T>(s->en->ep.kind, &ta, nullptr, s->en->ep, s->sh.E, s->stream));
        TRY(dispatch_step<T>(s->en->ep.kind, nullptr, &da, s->en->ep, s->sh.E, s->stream));
      } else if (s->en->is_user()) {
        if constexpr (sizeof(T) == 8) TRY(user_launch_jump(s->en, a, s->stream));
      } else {
        TRY(dispatch_jump<T>(s->en->ep.kind, a, s->en->ep, s->sh.E, s->stream));
      }
    }
    if (n_parts > 1 && i + 1 == n_iter) {  // the read-back follows every part
      for (int k = 1; k < n_parts; ++k) {
        HIPCHK(hipEventRecord(s->part_events[(size_t)k - 1], part_stream(k)));
        HIPCHK(hipStreamWaitEvent(s->stream, s->part_events[(size_t)k - 1], 0));
      }
    }
    HIPCHK(hipGetLastError());
    // sample download (mjhmc_iterate_download): iteration j is complete when every stream that ran part of it gets here --
    // and each of them re-tiles its own columns of the new X into staging slot j on the way (0.3 ms of a 17 ms C3
    // iteration): on a stream of its own that kernel waited for a workgroup of the NEXT iteration's persistent grid to
    // leave, so every slot crossed PCIe one iteration late and two of them after the run had ended.  The worker then only
    // moves bytes (copy engine + host).
    auto dl_done = [&](int j, const void* xj) -> int {
      if (!(s->dl && !s->dl->off && ring_slot0 >= 0 && j >= s->dl->marked.load(std::memory_order_acquire))) return 0;
      hipStream_t sts[kMaxDenseParts];
      const size_t elems = (size_t)s->D * s->N;
      const bool room = s->dl_stage_elems >= (size_t)s->dl->n_iter * elems;
      for (int k = 0; k < n_parts; ++k) {
        sts[k] = part_stream(k);
        if (room) TRY(dl_retile(s, xj, n_parts > 1 ? part_start(k) : 0, n_parts > 1 ? part_count(k) : s->N,
                                s->dl_stage + (size_t)j * elems, sts[k]));
      }
      s->dl->retiled[(size_t)j] = room ? 1 : 0;
      return dl_mark(s, j, 1, sts, n_parts);
    };
    TRY(dl_done(i, xo));
    xin = xo;
  }
  if (s->timing_on) HIPCHK(hipEventRecord(s->ev_total[1], s->stream));
  Control hc;
  std::vector<long long> hs((size_t)n_iter * 4);
  std::vector<int> hcnt((size_t)n_iter + 1);   // (+ the list the last iteration's movers went to)
  TRY(read_back_call(s, (size_t)n_iter, &hc, hs.data(), hcnt.data(), n_iter + 1));
  if (compact && carried) hcnt[0] = s->list_count;

  if (hc.failed && n_parts > 1 && n_iter > 1) {
    // a non-finite rate somewhere in the free-running parts: back to the state the call started from, and once more
    // on one stream -- that run stops at the failing iteration with the state, tallies and RNG position of a call that
    // was never split (nothing of this attempt has been committed: parities, Xcur and the tick are still the call's)
    TRY(split_state_copy(s, true));
    HIPCHK(hipStreamSynchronize(s->stream));
    TRY(dl_restart(s));   // (mjhmc_iterate_download: nothing the first run handed to the download survives it)
    return iterate_t<T>(s, n_iter, replay_normal, replay_exp, replay_unif, ring_slot0, per_iter, n_done, false);
  }

  const int done = hc.failed ? hc.failed_iter : n_iter;
  const int attempts = hc.failed ? done + 1 : n_iter;
  if (compact) {  // the cold tallies are the list lengths (of every part)
    for (int i = 0; i < attempts; ++i) hs[4 * (size_t)i + 3] = hcnt[(size_t)i];
  }
  fill_iter_stats(s, hs, attempts, done, hc.failed != 0, per_iter);
  if (compact && !hc.failed) {   // the next call's list
    s->list_par = ((carried ? s->list_par : 0) + n_iter) & 1;
    s->list_count = hcnt[(size_t)n_iter];
    s->list_valid = true;
  }
  // commit the finished iterations
  s->undo_valid = (n_iter == 1 && done == 1);  // the input buffers of a single iteration survive it: see mjhmc_rollback
  s->undo_multipass = false;
  s->undo_X = s->Xcur;
  if (done > 0) s->Xcur = xout[done - 1];
  s->vcur = (s->vcur + done) & 1;
  s->scur = (s->scur + done) & 1;
  s->tick += (uint64_t)attempts;
  if (n_done) *n_done = done;

  // One HIP-event pair brackets the whole launch sequence of this call (first jump kernel .. last jump
  // kernel, on the sampler's stream).  Per-launch event pairs were measured to cost more than they tell:
  // every marker packet opens a ~5 us bubble between back-to-back kernels.
  s->last_jump_launches = attempts;
  s->timing_pending = s->timing_on;
  return 0;
}

template <typename T>
static int leap_t(mjhmc_sampler& w, void* Xd, void* Vd, void* Xo, void* Vo, void* Gd, void* EXd, void* EVd, double eps,
                  int L) {
  LeapArgs<T> a;
  a.X = (const T*)Xd;
  a.V = (const T*)Vd;
  a.X_out = (T*)Xo;
  a.V_out = (T*)Vo;
  a.G = (T*)Gd;
  a.EX = (T*)EXd;
  a.EV = (T*)EVd;
  a.N = w.N;
  a.D = w.D;
  a.pitch = w.sh.pitch;
  a.CH = w.sh.CH;
  a.logG = w.sh.logG;
  a.L = L;
  a.eps = (T)eps;
  a.chalf = (T)(-eps / 2.);
  if constexpr (sizeof(T) == 8) {
    if (w.en->is_user()) return user_launch_leap(w.en, a, w.stream);
  }
  TRY(dispatch_leap<T>(w.en->ep.kind, a, w.en->ep, w.sh.E, w.stream));
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" {

int mjhmc_iterate(mjhmc_sampler* s, int n_iter, const double* replay_normal, const double* replay_exp,
                  const double* replay_unif, int ring_slot0, mjhmc_iter_stats* per_iter, int* n_done) {
  if (!s) return fail(MJHMC_ERR_INVALID, "sampler is NULL");
  if (n_iter < 1) return fail(MJHMC_ERR_INVALID, "n_iter must be >= 1");
  if (s->mode == MJHMC_MODE_CONTROL) {
    if ((replay_normal == nullptr) != (replay_unif == nullptr) || replay_exp)
      return fail(MJHMC_ERR_INVALID, "CONTROL replay takes replay_normal and replay_unif together");
  } else if ((replay_normal == nullptr) != (replay_exp == nullptr) || replay_unif) {
    return fail(MJHMC_ERR_INVALID, "jump-process replay takes replay_normal and replay_exp together");
  }
  if (ring_slot0 >= 0 && (!s->ring || ring_slot0 + n_iter > s->ring_slots))
    return fail(MJHMC_ERR_INVALID, "ring slots out of range (call mjhmc_ring_alloc)");
  if (s->en->is_host())
    return fail(MJHMC_ERR_UNSUPPORTED, "a host-evaluated energy is driven step by step: mjhmc_traj_begin / _step / _finish");
  HIPCHK(hipSetDevice(s->ctx->device));
  // ProductOfT with float64 state: the tile kernel with the state streamed through its epilogue (dense_pot64.hip); its
  // multi-pass form (the same arithmetic, bit for bit) serves L = 0 and, in the test build, the A/B switch
  const bool pot64_fused = s->sh.wide && s->en->is_pot() && !s->en->pot_big() && !s->sh.round32 && s->L >= 1 &&
                           !test_env("MJHMC_POT64_MULTIPASS");
  if (s->sh.wide && !pot64_fused) {
    s->list_valid = false;
    const int rc = multipass_iterate(s, n_iter, replay_normal, replay_exp, replay_unif, ring_slot0, per_iter, n_done);
    if (rc == 0) TRY(drop_spec(s));   // (that path neither reads nor writes the F-movers' hand-over: nothing of it is valid afterwards)
    return rc;
  }
  return s->dtype == MJHMC_F64
             ? iterate_t<double>(s, n_iter, replay_normal, replay_exp, replay_unif, ring_slot0, per_iter, n_done)
             : iterate_t<float>(s, n_iter, replay_normal, replay_exp, replay_unif, ring_slot0, per_iter, n_done);
}

int mjhmc_iterate_download(mjhmc_sampler* s, int n_iter, int ring_slot0, double* host_out, int64_t n_total, int64_t k0,
                           mjhmc_iter_stats* per_iter, int* n_done) {
  if (!s || !host_out || n_iter < 1) return fail(MJHMC_ERR_INVALID, "bad argument");
  if (ring_slot0 < 0 || !s->ring || ring_slot0 + n_iter > s->ring_slots)
    return fail(MJHMC_ERR_INVALID, "ring slots out of range (call mjhmc_ring_alloc)");
  if (k0 < 0 || k0 + n_iter > n_total) return fail(MJHMC_ERR_INVALID, "time indices [k0, k0 + n_iter) outside the host array");
  HIPCHK(hipSetDevice(s->ctx->device));
  if (!s->dl_stream) {
    // the re-tile kernel of a finished slot should get the first CUs a finishing workgroup of the persistent grids gives up
    int lo = 0, hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPCHK(hipStreamCreateWithPriority(&s->dl_stream, hipStreamNonBlocking, hi));
  }
  const size_t elems = (size_t)s->D * s->N;
  // staging in the host layout: a slot per iteration of the call where the device has the room (the iterations then
  // re-tile their own output, see iterate_t), else one slot the worker re-tiles into
  if (s->dl_stage_elems < (size_t)n_iter * elems) {
    double* big = nullptr;
    if (hipMalloc((void**)&big, (size_t)n_iter * elems * sizeof(double)) == hipSuccess) {
      if (s->dl_stage) HIPCHK(hipFree(s->dl_stage));
      s->dl_stage = big;
      s->dl_stage_elems = (size_t)n_iter * elems;
    } else {
      (void)hipGetLastError();
      if (s->dl_stage_elems < elems) {
        if (s->dl_stage) HIPCHK(hipFree(s->dl_stage));
        s->dl_stage = nullptr;
        s->dl_stage_elems = 0;
        HIPCHK(hipMalloc((void**)&s->dl_stage, elems * sizeof(double)));
        s->dl_stage_elems = elems;
      }
    }
  }
  for (int i = 0; i < 2; ++i) {
    if (!s->dl_pin[i]) HIPCHK(hipHostMalloc(&s->dl_pin[i], kPipeChunk, hipHostMallocDefault));
    if (!s->dl_ev[i]) HIPCHK(hipEventCreateWithFlags(&s->dl_ev[i], hipEventDisableTiming));
  }
  DlSession d;
  d.ev.resize((size_t)n_iter);
  d.retiled.assign((size_t)n_iter, 0);
  d.n_iter = n_iter;
  d.ring_slot0 = ring_slot0;
  d.host = host_out;
  d.n_total = n_total;
  d.k0 = k0;
  s->dl = &d;
  std::thread worker(dl_worker, s, &d);
  int done = 0;
  const int rc = mjhmc_iterate(s, n_iter, nullptr, nullptr, nullptr, ring_slot0, per_iter, &done);
  d.launched.store(true, std::memory_order_release);
  worker.join();
  s->dl = nullptr;
  int rc2 = d.rc;
  if (rc2) g_err = d.err;
  // what the worker did not bring over (a launch path that marks nothing; a re-run after a failure): plainly, now
  if (!rc && !rc2) {
    HIPCHK(hipStreamSynchronize(s->dl_stream));
    for (int i = d.downloaded.load(); i < done && !rc2; ++i) rc2 = dl_download_slot(s, &d, i);
  }
  dl_destroy_events(&d);
  if (n_done) *n_done = done;
  return rc ? rc : rc2;
}

static int read_vec(mjhmc_sampler* s, const void* dev, void* host, size_t nbytes) {
  // state scalars are stored in the state dtype; the ABI hands out float64
  if (nbytes != (size_t)s->N * sizeof(double)) return fail(MJHMC_ERR_INVALID, "expected N float64");
  if (s->dtype == MJHMC_F64) {
    HIPCHK(hipMemcpyAsync(host, dev, nbytes, hipMemcpyDeviceToHost, s->stream));
  } else {
    TRY(ensure_stage(s, (size_t)s->N));
    hipLaunchKernelGGL(widen_vec<float>, dim3((unsigned)((s->N + 255) / 256)), dim3(256), 0, s->stream,
                       (const float*)dev, s->stage, s->N);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host, s->stage, nbytes, hipMemcpyDeviceToHost, s->stream));
  }
  HIPCHK(hipStreamSynchronize(s->stream));
  return 0;
}

int mjhmc_read(mjhmc_sampler* s, int field, void* host_dst, size_t nbytes) {
  if (!s || !host_dst) return fail(MJHMC_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  const size_t mat = (size_t)s->D * s->N;
  switch (field) {
    case MJHMC_F_X:
    case MJHMC_F_V:
    case MJHMC_F_DEDX: {
      if (nbytes != mat * sizeof(double)) return fail(MJHMC_ERR_INVALID, "expected (D,N) float64");
      TRY(ensure_stage(s, mat));
      const void* src = field == MJHMC_F_X ? s->Xcur : s->Vbuf[s->vcur];
      if (field == MJHMC_F_DEDX && (s->en->is_pot() || s->en->is_host() || s->sh.wide)) {
        src = s->Gbuf[s->vcur];
      } else if (field == MJHMC_F_DEDX && s->en->is_sic()) {
        if (!s->scratch) HIPCHK(hipMalloc(&s->scratch, (size_t)s->Npad * s->D * sizeof(float)));
        TRY(run_eval(s, s->Xcur, s->scratch, nullptr, nullptr, nullptr, nullptr));
        s->download_f32 = true;
        const int rc = download_cols(s, s->scratch, nullptr, s->N, (double*)host_dst, mat, s->N, 1, 0, true);
        s->download_f32 = false;
        return rc;
      } else if (field == MJHMC_F_DEDX) {
        if (!s->scratch) HIPCHK(hipMalloc(&s->scratch, mat_bytes(s)));
        TRY(run_eval(s, s->Xcur, s->scratch, nullptr, nullptr, nullptr, nullptr));
        src = s->scratch;
      }
      return download_cols(s, src, nullptr, s->N, (double*)host_dst, mat, s->N, 1, 0, true);
    }
    case MJHMC_F_EX: return read_vec(s, s->EX[s->scur], host_dst, nbytes);
    case MJHMC_F_EV: return read_vec(s, s->EV[s->scur], host_dst, nbytes);
    case MJHMC_F_HFLF: return read_vec(s, s->Hflf[s->scur], host_dst, nbytes);
    case MJHMC_F_DWELL:
      if (nbytes != (size_t)s->N * sizeof(double)) return fail(MJHMC_ERR_INVALID, "expected N float64");
      HIPCHK(hipMemcpyAsync(host_dst, s->dwell, nbytes, hipMemcpyDeviceToHost, s->stream));
      HIPCHK(hipStreamSynchronize(s->stream));
      return 0;
    case MJHMC_F_CACHE: {
      if (nbytes != (size_t)s->N) return fail(MJHMC_ERR_INVALID, "expected N uint8");
      TRY(ensure_stage(s, (size_t)(s->N + 7) / 8));
      uint8_t* flags = (uint8_t*)s->stage;
      const dim3 g((unsigned)((s->N + 255) / 256)), b(256);
      if (s->dtype == MJHMC_F64)
        hipLaunchKernelGGL(flags_from_hflf<double>, g, b, 0, s->stream, (const double*)s->Hflf[s->scur], flags, s->N);
      else
        hipLaunchKernelGGL(flags_from_hflf<float>, g, b, 0, s->stream, (const float*)s->Hflf[s->scur], flags, s->N);
      HIPCHK(hipGetLastError());
      HIPCHK(hipMemcpyAsync(host_dst, flags, nbytes, hipMemcpyDeviceToHost, s->stream));
      HIPCHK(hipStreamSynchronize(s->stream));
      return 0;
    }
    case MJHMC_F_TRANS:
      if (nbytes != (size_t)s->N) return fail(MJHMC_ERR_INVALID, "expected N uint8");
      HIPCHK(hipMemcpyAsync(host_dst, s->trans, nbytes, hipMemcpyDeviceToHost, s->stream));
      HIPCHK(hipStreamSynchronize(s->stream));
      return 0;
    default: return fail(MJHMC_ERR_INVALID, "unknown field");
  }
}

int mjhmc_write(mjhmc_sampler* s, int field, const void* host_src, size_t nbytes) {
  if (!s || !host_src) return fail(MJHMC_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  const size_t mat = (size_t)s->D * s->N;
  switch (field) {
    case MJHMC_F_X:
    case MJHMC_F_V: {
      if (nbytes != mat * sizeof(double)) return fail(MJHMC_ERR_INVALID, "expected (D,N) float64");
      void* dst = field == MJHMC_F_X ? s->Xcur : s->Vbuf[s->vcur];
      s->undo_valid = false;
      s->list_valid = false;
      if (field == MJHMC_F_X) s->host_energy_set = false;  // MJHMC_E_HOST: E and dE/dX of the new X are the caller's to supply
      TRY(upload_matrix(s, (const double*)host_src, dst));
      if (s->sh.round32) TRY(round_rows32(s, dst));
      TRY(run_eval(s, s->Xcur, s->Gbuf[s->vcur], s->EX[s->scur], s->Vbuf[s->vcur], nullptr, s->EV[s->scur]));
      HIPCHK(hipMemsetAsync(s->Hflf[s->scur], 0xFF, s->Npad * ssize(s), s->stream));
      TRY(drop_spec(s));
      HIPCHK(hipStreamSynchronize(s->stream));
      return 0;
    }
    case MJHMC_F_HFLF: {  // float64 (N); NaN marks a cold cache entry
      if (nbytes != (size_t)s->N * sizeof(double)) return fail(MJHMC_ERR_INVALID, "expected N float64");
      TRY(drop_spec(s));
      s->list_valid = false;
      if (s->dtype == MJHMC_F64) {
        HIPCHK(hipMemcpyAsync(s->Hflf[s->scur], host_src, nbytes, hipMemcpyHostToDevice, s->stream));
      } else {
        TRY(ensure_stage(s, (size_t)s->N));
        HIPCHK(hipMemcpyAsync(s->stage, host_src, nbytes, hipMemcpyHostToDevice, s->stream));
        hipLaunchKernelGGL(narrow_vec<float>, dim3((unsigned)((s->N + 255) / 256)), dim3(256), 0, s->stream,
                           (const double*)s->stage, (float*)s->Hflf[s->scur], s->N);
        HIPCHK(hipGetLastError());
      }
      HIPCHK(hipStreamSynchronize(s->stream));
      return 0;
    }
    default: return fail(MJHMC_ERR_INVALID, "field is not writable");
  }
}

int mjhmc_ring_alloc(mjhmc_sampler* s, int n_slots) {
  if (!s || n_slots < 1) return fail(MJHMC_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  if (n_slots <= s->ring_slots) return 0;
  HIPCHK(hipStreamSynchronize(s->stream));
  const size_t mb = mat_bytes(s);
  // the NEW ring first: a request the device cannot hold leaves the sampler with the ring it had
  void* ring = nullptr;
  double* dring = nullptr;
  hipError_t e = hipMalloc(&ring, (size_t)n_slots * mb);
  if (e == hipSuccess) e = hipMalloc((void**)&dring, (size_t)n_slots * s->Npad * sizeof(double));
  if (e != hipSuccess) {
    if (ring) (void)hipFree(ring);
    (void)hipGetLastError();
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    char msg[320];
    std::snprintf(msg, sizeof(msg), "a sample ring of %d slots x %.3f GB = %.1f GB does not fit the device (%.1f GB free of %.1f GB): "
                  "record in chunks (mjhmc_iterate_download over a smaller ring) -- the ring the sampler had is kept",
                  n_slots, mb / 1e9, (double)n_slots * mb / 1e9, free_b / 1e9, total_b / 1e9);
    return fail(MJHMC_ERR_HIP, msg);
  }
  // (on the sampler's stream: it does not wait for the null stream, and a memset there can land AFTER the first slots are written)
  HIPCHK(hipMemsetAsync(ring, 0, (size_t)n_slots * mb, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  s->undo_valid = false;
  // the live X may sit in the old ring: park it in a ping-pong buffer before freeing
  if (s->ring && (char*)s->Xcur >= (char*)s->ring && (char*)s->Xcur < (char*)s->ring + (size_t)s->ring_slots * mb) {
    HIPCHK(hipMemcpy(s->Xbuf[0], s->Xcur, mb, hipMemcpyDeviceToDevice));
    s->Xcur = s->Xbuf[0];
  }
  if (s->ring) HIPCHK(hipFree(s->ring));
  if (s->dwell_ring) HIPCHK(hipFree(s->dwell_ring));
  s->ring = ring;
  s->dwell_ring = dring;
  s->ring_slots = n_slots;
  return 0;
}

int mjhmc_ring_slot_bytes(mjhmc_sampler* s, uint64_t* bytes) {
  if (!s || !bytes) return fail(MJHMC_ERR_INVALID, "NULL argument");
  *bytes = (uint64_t)mat_bytes(s) + (uint64_t)s->Npad * sizeof(double);
  return 0;
}

int mjhmc_mem_info(mjhmc_ctx* ctx, uint64_t* free_bytes, uint64_t* total_bytes) {
  if (!ctx || !free_bytes || !total_bytes) return fail(MJHMC_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(ctx->device));
  size_t f = 0, t = 0;
  HIPCHK(hipMemGetInfo(&f, &t));
  *free_bytes = f;
  *total_bytes = t;
  return 0;
}

int mjhmc_ring_read_dwell(mjhmc_sampler* s, int slot0, int n, double* host_dst) {
  if (!s || !host_dst) return fail(MJHMC_ERR_INVALID, "NULL argument");
  if (slot0 < 0 || n < 1 || slot0 + n > s->ring_slots) return fail(MJHMC_ERR_INVALID, "slots out of range");
  HIPCHK(hipSetDevice(s->ctx->device));
  HIPCHK(hipMemcpy2DAsync(host_dst, (size_t)s->N * sizeof(double), s->dwell_ring + (size_t)slot0 * s->Npad,
                          (size_t)s->Npad * sizeof(double), (size_t)s->N * sizeof(double), (size_t)n,
                          hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  return 0;
}

int mjhmc_ring_gather(mjhmc_sampler* s, const int64_t* idx, int64_t n, double* host_out) {
  if (!s || !idx || !host_out || n < 1) return fail(MJHMC_ERR_INVALID, "bad argument");
  HIPCHK(hipSetDevice(s->ctx->device));
  const int64_t pool = (int64_t)s->ring_slots * s->N;
  for (int64_t k = 0; k < n; ++k)
    if (idx[k] < 0 || idx[k] >= pool) return fail(MJHMC_ERR_INVALID, "gather index outside the sample ring");
  std::vector<int64_t> rows((size_t)n);  // pool index (slot * N + p) -> padded row (slot * Npad + p)
  for (int64_t k = 0; k < n; ++k) rows[k] = (idx[k] / s->N) * s->Npad + idx[k] % s->N;
  int64_t* didx = nullptr;
  HIPCHK(hipMalloc((void**)&didx, n * sizeof(int64_t)));
  int rc = 0;
  do {
    if (hipMemcpyAsync(didx, rows.data(), n * sizeof(int64_t), hipMemcpyHostToDevice, s->stream) != hipSuccess) {
      rc = fail(MJHMC_ERR_HIP, "index upload failed");
      break;
    }
    rc = ensure_stage(s, (size_t)s->D * n);
    if (rc) break;
    rc = download_cols(s, s->ring, didx, n, host_out, (size_t)s->D * n, n, 1, 0, true);
  } while (0);
  (void)hipFree(didx);
  return rc;
}

int mjhmc_ring_read(mjhmc_sampler* s, int slot0, int n, int stacked, double* host_out) {
  if (!s || !host_out) return fail(MJHMC_ERR_INVALID, "NULL argument");
  if (slot0 < 0 || n < 1 || slot0 + n > s->ring_slots) return fail(MJHMC_ERR_INVALID, "slots out of range");
  HIPCHK(hipSetDevice(s->ctx->device));
  const size_t total = (size_t)s->D * s->N * n;
  TRY(ensure_stage(s, total));
  const size_t mb = mat_bytes(s);
  const char* base = (const char*)s->ring + (size_t)slot0 * mb;
  if (!stacked) {
    for (int k = 0; k < n; ++k)
      TRY(download_cols(s, base + (size_t)k * mb, nullptr, s->N, host_out, total, (int64_t)n * s->N, 1,
                        (int64_t)k * s->N, k == n - 1));
  } else {
    for (int k = 0; k < n; ++k)
      TRY(download_cols(s, base + (size_t)k * mb, nullptr, s->N, host_out, total, (int64_t)s->N * n, n, k, k == n - 1));
  }
  return 0;
}

int mjhmc_ring_moments(mjhmc_sampler* s, int slot0, int n, double shift, double* sum, double* sumsq) {
  if (!s || !sum || !sumsq) return fail(MJHMC_ERR_INVALID, "NULL argument");
  if (slot0 < 0 || n < 1 || slot0 + n > s->ring_slots) return fail(MJHMC_ERR_INVALID, "slots out of range");
  HIPCHK(hipSetDevice(s->ctx->device));
  TRY(ensure_stage(s, 2));
  HIPCHK(hipMemsetAsync(s->stage, 0, 2 * sizeof(double), s->stream));
  const char* base = (const char*)s->ring + (size_t)slot0 * mat_bytes(s);
  const dim3 grid(1024), block(256);
  if (s->dtype == MJHMC_F64)
    hipLaunchKernelGGL(moments_kernel<double>, grid, block, 0, s->stream, (const double*)base, s->Npad, s->N, n, s->D,
                       s->sh.pitch, shift, s->stage);
  else if (s->dtype == MJHMC_BF16)
    hipLaunchKernelGGL(moments_kernel<__bf16>, grid, block, 0, s->stream, (const __bf16*)base, s->Npad, s->N, n, s->D,
                       s->sh.pitch, shift, s->stage);
  else
    hipLaunchKernelGGL(moments_kernel<float>, grid, block, 0, s->stream, (const float*)base, s->Npad, s->N, n, s->D,
                       s->sh.pitch, shift, s->stage);
  HIPCHK(hipGetLastError());
  double h[2];
  HIPCHK(hipMemcpyAsync(h, s->stage, sizeof(h), hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  *sum = h[0];
  *sumsq = h[1];
  return 0;
}

int mjhmc_ring_autocor(mjhmc_sampler* s, int slot0, int n, int linear, double* host_out) {
  if (!s || !host_out) return fail(MJHMC_ERR_INVALID, "NULL argument");
  if (slot0 < 0 || n < 1 || slot0 + n > s->ring_slots) return fail(MJHMC_ERR_INVALID, "slots out of range");
  HIPCHK(hipSetDevice(s->ctx->device));
  const RingView view{(const char*)s->ring + (size_t)slot0 * mat_bytes(s), s->dtype, s->Npad, s->N, s->D, s->sh.pitch};
  std::string err;
  const int rc = autocor_from_ring(s->stream, view, n, linear, host_out, err);
  return rc ? fail(rc, err) : 0;
}

int mjhmc_draw_from(mjhmc_ctx* ctx, const double* rates, const double* unit_exp, int64_t n, double* out,
                    int64_t* first_bad) {
  if (!ctx || !rates || !unit_exp || !out || n < 0) return fail(MJHMC_ERR_INVALID, "bad argument");
  if (first_bad) *first_bad = -1;
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  double *d_r = nullptr, *d_e = nullptr, *d_o = nullptr;
  long long* d_bad = nullptr;
  const size_t bytes = (size_t)n * sizeof(double);
  const long long none = 0x7fffffffffffffffLL;
  long long bad = none;
  hipError_t rc = hipMalloc(&d_r, 3 * bytes + sizeof(long long));
  if (rc != hipSuccess) return fail(MJHMC_ERR_HIP, hipGetErrorString(rc));
  d_e = d_r + n;
  d_o = d_e + n;
  d_bad = reinterpret_cast<long long*>(d_o + n);
  rc = hipMemcpy(d_r, rates, bytes, hipMemcpyHostToDevice);
  if (rc == hipSuccess) rc = hipMemcpy(d_e, unit_exp, bytes, hipMemcpyHostToDevice);
  if (rc == hipSuccess) rc = hipMemcpy(d_bad, &none, sizeof(long long), hipMemcpyHostToDevice);
  if (rc == hipSuccess) {
    hipLaunchKernelGGL(draw_from_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_r, d_e, n, d_o, d_bad);
    rc = hipGetLastError();
  }
  if (rc == hipSuccess) rc = hipMemcpy(out, d_o, bytes, hipMemcpyDeviceToHost);
  if (rc == hipSuccess) rc = hipMemcpy(&bad, d_bad, sizeof(long long), hipMemcpyDeviceToHost);
  (void)hipFree(d_r);
  if (rc != hipSuccess) return fail(MJHMC_ERR_HIP, hipGetErrorString(rc));
  if (bad != none) {
    if (first_bad) *first_bad = (int64_t)bad;
    return fail(MJHMC_ERR_NONFINITE, "Infinite rate (mjhmc/misc/utils.py:43-48)");
  }
  return 0;
}

int mjhmc_min_idx(mjhmc_ctx* ctx, const double* draws, int k, int64_t n, int32_t* which) {
  if (!ctx || !draws || !which || k < 1 || n < 0) return fail(MJHMC_ERR_INVALID, "bad argument");
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(ctx->device));
  double* d_d = nullptr;
  const size_t bytes = (size_t)k * (size_t)n * sizeof(double);
  hipError_t rc = hipMalloc(&d_d, bytes + (size_t)n * sizeof(int32_t));
  if (rc != hipSuccess) return fail(MJHMC_ERR_HIP, hipGetErrorString(rc));
  int32_t* d_w = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(d_d) + bytes);
  rc = hipMemcpy(d_d, draws, bytes, hipMemcpyHostToDevice);
  if (rc == hipSuccess) {
    hipLaunchKernelGGL(min_idx_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_d, k, n, d_w);
    rc = hipGetLastError();
  }
  if (rc == hipSuccess) rc = hipMemcpy(which, d_w, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost);
  (void)hipFree(d_d);
  if (rc != hipSuccess) return fail(MJHMC_ERR_HIP, hipGetErrorString(rc));
  return 0;
}

int mjhmc_autocor(mjhmc_ctx* ctx, const double* samples, int64_t n_series, int n_samples, int linear,
                  double* host_out) {
  if (!ctx || !samples || !host_out) return fail(MJHMC_ERR_INVALID, "NULL argument");
  HIPCHK(hipSetDevice(ctx->device));
  std::string err;
  const int rc = autocor_from_host(nullptr, samples, n_series, n_samples, linear, host_out, err);
  return rc ? fail(rc, err) : 0;
}

int mjhmc_last_timing(mjhmc_sampler* s, double* total_ms, double* jump_kernel_ms, int* n_jump_launches) {
  if (!s) return fail(MJHMC_ERR_INVALID, "sampler is NULL");
  if (s->timing_pending) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, s->ev_total[0], s->ev_total[1]));
    s->last_total_ms = s->last_jump_ms = ms;
    s->timing_pending = false;
  }
  if (total_ms) *total_ms = s->last_total_ms;
  if (jump_kernel_ms) *jump_kernel_ms = s->last_jump_ms;
  if (n_jump_launches) *n_jump_launches = s->last_jump_launches;
  return 0;
}

int mjhmc_set_timing(mjhmc_sampler* s, int on) {
  if (!s) return fail(MJHMC_ERR_INVALID, "sampler is NULL");
  s->timing_on = on != 0;
  if (!s->timing_on) s->timing_pending = false;
  return 0;
}

int mjhmc_sync(mjhmc_sampler* s) {
  if (!s) return fail(MJHMC_ERR_INVALID, "sampler is NULL");
  HIPCHK(hipSetDevice(s->ctx->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  return 0;
}

int mjhmc_eval(mjhmc_energy* e, int dtype, const double* X, int64_t n, double* E_out, double* dEdX_out) {
  if (!e || !X || n < 1) return fail(MJHMC_ERR_INVALID, "bad argument");
  if (dtype != MJHMC_F64 && dtype != MJHMC_F32 && dtype != MJHMC_BF16)
    return fail(MJHMC_ERR_INVALID, "dtype must be F64, F32 or BF16");
  if ((dtype == MJHMC_BF16 && !e->is_sic()) || (e->is_sic() && dtype == MJHMC_F64))
    return fail(MJHMC_ERR_UNSUPPORTED, "SPARSE_CODE evaluates with BF16 or F32 state (and BF16 state is its alone)");
  if (e->is_user() && dtype != MJHMC_F64) return fail(MJHMC_ERR_UNSUPPORTED, "user-expression energies run in float64");
  if (e->is_host()) return fail(MJHMC_ERR_UNSUPPORTED, "a host-evaluated energy has no device evaluation: call the callables");
  HIPCHK(hipSetDevice(e->ctx->device));
  // a throw-away sampler-shaped workspace keeps one code path for re-tiling and evaluation
  mjhmc_sampler w;
  w.ctx = e->ctx;
  w.en = e;
  w.N = n;
  w.Npad = (n + 63) / 64 * 64;
  w.first_pid = 0;
  w.D = e->ep.ndims;
  w.dtype = dtype;
  TRY(shape_for(e, &w.dtype, &w.sh));
  dtype = w.dtype;
  void *Xd = nullptr, *Gd = nullptr, *Ed = nullptr;
  auto body = [&]() -> int {
    HIPCHK(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    const size_t mb = mat_bytes(&w);
    HIPCHK(hipMalloc(&Xd, mb));
    HIPCHK(hipMemsetAsync(Xd, 0, mb, w.stream));
    if (dEdX_out) HIPCHK(hipMalloc(&Gd, e->is_sic() ? (size_t)w.Npad * w.D * sizeof(float) : mb));
    if (E_out) HIPCHK(hipMalloc(&Ed, w.Npad * ssize(&w)));
    TRY(upload_matrix(&w, X, Xd));
    if (w.sh.round32) TRY(round_rows32(&w, Xd));
    TRY(run_eval(&w, Xd, Gd, Ed, nullptr, nullptr, nullptr));
    if (E_out) TRY(read_vec(&w, Ed, E_out, (size_t)n * sizeof(double)));
    w.download_f32 = e->is_sic();
    if (dEdX_out) TRY(download_cols(&w, Gd, nullptr, n, dEdX_out, (size_t)w.D * n, n, 1, 0, true));
    HIPCHK(hipStreamSynchronize(w.stream));
    return 0;
  };
  const int rc = body();
  if (Xd) (void)hipFree(Xd);
  if (Gd) (void)hipFree(Gd);
  if (Ed) (void)hipFree(Ed);
  if (w.stage) (void)hipFree(w.stage);
  for (int i = 0; i < 2; ++i) {  // pinned buffers of copy_to_host, if a result was big enough to take that path
    if (w.pipe_pin[i]) (void)hipHostFree(w.pipe_pin[i]);
    if (w.pipe_ev[i]) (void)hipEventDestroy(w.pipe_ev[i]);
  }
  if (w.stream) (void)hipStreamDestroy(w.stream);
  return rc;
}

int mjhmc_leapfrog(mjhmc_energy* e, int dtype, const double* X, const double* V, int64_t n, double eps, int n_steps,
                   double* X_out, double* V_out, double* EX_out, double* EV_out, double* dEdX_out) {
  if (!e || !X || !V || !X_out || !V_out || n < 1 || n_steps < 0) return fail(MJHMC_ERR_INVALID, "bad argument");
  if (dtype != MJHMC_F64 && dtype != MJHMC_F32 && dtype != MJHMC_BF16)
    return fail(MJHMC_ERR_INVALID, "dtype must be F64, F32 or BF16");
  if ((dtype == MJHMC_BF16 && !e->is_sic()) || (e->is_sic() && dtype == MJHMC_F64))
    return fail(MJHMC_ERR_UNSUPPORTED, "SPARSE_CODE runs with BF16 or F32 state (and BF16 state is its alone)");
  if (e->is_user() && dtype != MJHMC_F64) return fail(MJHMC_ERR_UNSUPPORTED, "user-expression energies run in float64");
  if (e->is_host()) return fail(MJHMC_ERR_UNSUPPORTED, "a host-evaluated energy has no device leapfrog operator");
  HIPCHK(hipSetDevice(e->ctx->device));
  mjhmc_sampler w;
  w.ctx = e->ctx;
  w.en = e;
  w.N = n;
  w.Npad = (n + 63) / 64 * 64;
  w.first_pid = 0;
  w.D = e->ep.ndims;
  w.dtype = dtype;
  TRY(shape_for(e, &w.dtype, &w.sh));
  dtype = w.dtype;
  void* buf[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // X, V, X', V', G, EX, EV
  auto body = [&]() -> int {
    HIPCHK(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    const size_t mb = mat_bytes(&w), vb = (size_t)w.Npad * ssize(&w);
    const size_t gb = e->is_sic() ? (size_t)w.Npad * w.D * sizeof(float) : mb;  // SPARSE_CODE hands dE/dX out in float32
    for (int i = 0; i < 5; ++i) {
      if (i == 4 && !dEdX_out) continue;
      HIPCHK(hipMalloc(&buf[i], i == 4 ? gb : mb));
      HIPCHK(hipMemsetAsync(buf[i], 0, i == 4 ? gb : mb, w.stream));
    }
    if (EX_out) HIPCHK(hipMalloc(&buf[5], vb));
    if (EV_out) HIPCHK(hipMalloc(&buf[6], vb));
    TRY(upload_matrix(&w, X, buf[0]));
    TRY(upload_matrix(&w, V, buf[1]));
    if (w.sh.round32) {
      TRY(round_rows32(&w, buf[0]));
      TRY(round_rows32(&w, buf[1]));
    }
    if (e->is_pot() && !w.sh.wide) {
      PotLeapArgs a;
      a.X = (const float*)buf[0];
      a.V = (const float*)buf[1];
      a.X_out = (float*)buf[2];
      a.V_out = (float*)buf[3];
      a.G = (float*)buf[4];
      a.EX = (float*)buf[5];
      a.EV = (float*)buf[6];
      a.N = n;
      a.ntiles = w.Npad / 32;
      a.D = w.D;
      a.L = n_steps;
      a.eps = (float)eps;
      a.chalf = (float)(-eps / 2.);
      pot_launch_leap(a, e->pot_model(), w.stream);
      HIPCHK(hipGetLastError());
    } else if (e->is_sic()) {
      auto leap_sic = [&](auto tag) {
        using ST = decltype(tag);
        SicLeapArgsT<ST> a;
        a.X = (const ST*)buf[0];
        a.V = (const ST*)buf[1];
        a.X_out = (ST*)buf[2];
        a.V_out = (ST*)buf[3];
        a.G = (float*)buf[4];
        a.EX = (float*)buf[5];
        a.EV = (float*)buf[6];
        a.N = n;
        const int ppt = sic_particles_per_tile(e->sic_P);
        a.ntiles = (n + ppt - 1) / ppt;
        a.L = n_steps;
        a.eps = (float)eps;
        a.chalf = (float)(-eps / 2.);
        sic_launch_leap(a, e->sic_model(), w.stream);
      };
      if (dtype == MJHMC_BF16) leap_sic(__bf16{});
      else leap_sic(float{});
      HIPCHK(hipGetLastError());
    } else if (w.sh.wide) {
      TRY(wide_leapfrog(&w, (const double*)buf[0], (const double*)buf[1], (double*)buf[2], (double*)buf[3], (double*)buf[4],
                        (double*)buf[5], (double*)buf[6], eps, n_steps));
    } else if (dtype == MJHMC_F64) {
      TRY(leap_t<double>(w, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], eps, n_steps));
    } else {
      TRY(leap_t<float>(w, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], eps, n_steps));
    }
    const size_t total = (size_t)w.D * n;
    TRY(download_cols(&w, buf[2], nullptr, n, X_out, total, n, 1, 0, true));
    TRY(download_cols(&w, buf[3], nullptr, n, V_out, total, n, 1, 0, true));
    if (dEdX_out) {
      w.download_f32 = e->is_sic();
      TRY(download_cols(&w, buf[4], nullptr, n, dEdX_out, total, n, 1, 0, true));
      w.download_f32 = false;
    }
    if (EX_out) TRY(read_vec(&w, buf[5], EX_out, (size_t)n * sizeof(double)));
    if (EV_out) TRY(read_vec(&w, buf[6], EV_out, (size_t)n * sizeof(double)));
    HIPCHK(hipStreamSynchronize(w.stream));
    return 0;
  };
  const int rc = body();
  for (void* b : buf)
    if (b) (void)hipFree(b);
  if (w.stage) (void)hipFree(w.stage);
  for (int i = 0; i < 2; ++i) {  // pinned buffers of copy_to_host, if a result was big enough to take that path
    if (w.pipe_pin[i]) (void)hipHostFree(w.pipe_pin[i]);
    if (w.pipe_ev[i]) (void)hipEventDestroy(w.pipe_ev[i]);
  }
  if (w.stream) (void)hipStreamDestroy(w.stream);
  return rc;
}

}  // extern "C"
