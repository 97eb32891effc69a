// Autocorrelation of the recorded samples (mjhmc/misc/autocor.py:37-49, :177-211).
//
// The reference transforms the host array [n_dims, n_batch, n_samples] along time, multiplies by the
// conjugate, transforms back every series and only then averages over dims and particles.  The inverse
// transform is linear, so the average moves in front of it: per chunk of series
//   ring (time-major) --gather_series--> [series][time] float64 --batched D2Z (hipFFT)--> spectra
//   --power_accumulate--> P[f] += |F_series[f]|^2
// and ONE length-T inverse transform of P at the end.  HBM traffic is one read of the ring plus the
// transform's own passes over a bounded staging buffer; nothing of size (D, N, T) goes to the host.
// hipFFT is loaded on first use (dlopen) so that the sampling path does not page in rocFFT.
#include "autocor.hpp"

#include <dlfcn.h>
#include <hipfft/hipfft.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "../../include/mjhmc_hip.h"

namespace {

struct FftApi {
  void* lib = nullptr;
  hipfftResult (*planMany)(hipfftHandle*, int, int*, int*, int, int, int*, int, int, hipfftType, int) = nullptr;
  hipfftResult (*setStream)(hipfftHandle, hipStream_t) = nullptr;
  hipfftResult (*execD2Z)(hipfftHandle, hipfftDoubleReal*, hipfftDoubleComplex*) = nullptr;
  hipfftResult (*execZ2D)(hipfftHandle, hipfftDoubleComplex*, hipfftDoubleReal*) = nullptr;
  hipfftResult (*destroy)(hipfftHandle) = nullptr;
};

template <typename F>
bool bind(void* lib, const char* name, F& fn) {
  fn = reinterpret_cast<F>(dlsym(lib, name));
  return fn != nullptr;
}

const FftApi* fft_api(std::string& err) {
  static FftApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"libhipfft.so.0", "libhipfft.so", "/opt/rocm/lib/libhipfft.so.0"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.lib) break;
    }
    if (api.lib) {
      const bool ok = bind(api.lib, "hipfftPlanMany", api.planMany) && bind(api.lib, "hipfftSetStream", api.setStream) &&
                      bind(api.lib, "hipfftExecD2Z", api.execD2Z) && bind(api.lib, "hipfftExecZ2D", api.execZ2D) &&
                      bind(api.lib, "hipfftDestroy", api.destroy);
      if (!ok) {
        dlclose(api.lib);
        api.lib = nullptr;
      }
    }
  }
  if (!api.lib) {
    err = "hipFFT (libhipfft.so.0) could not be loaded; the device autocorrelation has no other path";
    return nullptr;
  }
  return &api;
}

// 64 series x 64 time steps per block through LDS: reads run along the state row (coalesced over d),
// writes along time.  Series past `ns` and times past `nT` (zero padding of the linear variant) are zero.
template <typename T>
__global__ __launch_bounds__(256) void gather_series(const T* __restrict__ ring, size_t slot_elems, int pitch, int D,
                                                     int nT, int M, int64_t s0, int64_t ns,
                                                     double* __restrict__ out) {
  __shared__ double tile[64][65];
  const int t0 = blockIdx.y * 64;
  const int64_t l0 = (int64_t)blockIdx.x * 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t sl = l0 + lane;
  const bool live = sl < ns;
  const int64_t g = s0 + sl;
  const int64_t p = g / D;
  const int d = (int)(g - p * D);
  const size_t off = (size_t)p * pitch + d;
#pragma unroll 4
  for (int r = 0; r < 16; ++r) {
    const int tl = r * 4 + w, t = t0 + tl;
    double v = 0.0;
    if (live && t < nT) v = (double)ring[(size_t)t * slot_elems + off];
    tile[tl][lane] = v;
  }
  __syncthreads();
  const int t = t0 + lane;
  if (t < M) {
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
      const int sr = r * 4 + w;
      out[(size_t)(l0 + sr) * M + t] = tile[lane][sr];
    }
  }
}

// P[f] += sum over the chunk's series of |F[series][f]|^2 (P is complex with zero imaginary part: the
// input of the final inverse transform)
__global__ __launch_bounds__(256) void power_accumulate(const double2* __restrict__ F, int nF, int64_t rows,
                                                        double2* __restrict__ P) {
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= nF) return;
  double acc = 0.0;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const double2 z = F[(size_t)r * nF + f];
    acc += z.x * z.x + z.y * z.y;
  }
  atomicAdd(&P[f].x, acc);
}

// ---------------------------------------------------------------------------------------------------------------------
// Short series straight from the ring (round 6).  At the batch sizes of BASELINE.json a ring holds tens of samples, not
// thousands (C2: 410 MB per sample): for T <= 32 the transform pipeline above -- gather into [series][time], batched
// transform, power sums: ~5 passes over a staging copy of the ring -- costs ten times the ONE read of the ring that the
// lag products need (C2, 16 samples: 17.4 ms against 1.8 ms for recording them).  A thread owns a series (particle p,
// coordinate d: consecutive threads, consecutive coordinates of a state row -- coalesced), reads its T values (zeros up
// to the compile-time TT) and adds all LINEAR lag products x[t] x[t + k] into TT accumulators; a persistent grid, partial
// sums per block, a second small kernel adds the blocks in a fixed order (no atomics: the result does not depend on the
// schedule).  The circular sums of fft_autocor follow on the host: circ[k] = lin[k] + lin[T - k], circ[0] = lin[0].
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int TT>
__global__ __launch_bounds__(256) void lag_sums_direct(const T* __restrict__ ring, size_t slot_elems, int pitch, int D, int nT,
                                                       int64_t n_series, double* __restrict__ partial) {
  double acc[TT];
#pragma unroll
  for (int k = 0; k < TT; ++k) acc[k] = 0.0;
  const uint32_t uD = (uint32_t)D;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n_series; g += (int64_t)gridDim.x * 256) {
    // (n_series < 2^32 at every size a device holds: 32-bit division)
    const uint32_t p = (uint32_t)((uint64_t)g / uD), d = (uint32_t)((uint64_t)g - (uint64_t)p * uD);
    const size_t off = (size_t)p * pitch + d;
    double x[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) x[t] = t < nT ? (double)ring[(size_t)t * slot_elems + off] : 0.0;
#pragma unroll
    for (int k = 0; k < TT; ++k) {
      double s = 0.0;
#pragma unroll
      for (int t = 0; t + k < TT; ++t) s = __builtin_fma(x[t], x[t + k], s);
      acc[k] += s;
    }
  }
  __shared__ double red[4][TT];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < TT; ++k) {
    double v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) red[w][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < TT) partial[(size_t)blockIdx.x * TT + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

template <int TT>
__global__ void lag_sums_finish(const double* __restrict__ partial, int n_blocks, double* __restrict__ out) {
  const int k = threadIdx.x;
  if (k >= TT) return;
  double s = 0.0;
  for (int b = 0; b < n_blocks; ++b) s += partial[(size_t)b * TT + k];
  out[k] = s;
}

// The most recent pair of plans is kept per thread: building them costs ~10 ms (and ~1 s the first time,
// when rocFFT loads its kernels), a transform of a few GB of samples ~10 ms.
struct PlanPair {
  const FftApi* api = nullptr;
  hipfftHandle fwd = 0, inv = 0;
  bool fwd_live = false, inv_live = false;
  int device = -1, M = 0;
  int64_t chunk = 0;
  void release() {
    if (fwd_live) api->destroy(fwd);
    if (inv_live) api->destroy(inv);
    fwd_live = inv_live = false;
  }
  // no destructor: at process exit the HIP runtime may already be gone, and the OS reclaims the plans
};

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
};

#define ACHK(expr)                                                                                      \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) {                                                                             \
      err = std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + ":" +                 \
            std::to_string(__LINE__) + ")";                                                             \
      return MJHMC_ERR_HIP;                                                                             \
    }                                                                                                   \
  } while (0)
#define FCHK(expr)                                                                                      \
  do {                                                                                                  \
    hipfftResult r_ = (expr);                                                                           \
    if (r_ != HIPFFT_SUCCESS) {                                                                         \
      err = std::string(#expr) + ": hipfftResult " + std::to_string((int)r_);                           \
      return MJHMC_ERR_HIP;                                                                             \
    }                                                                                                   \
  } while (0)

// staging budget: series-major float64 input + spectra of one chunk
size_t staging_budget() {
#ifdef MJHMC_TEST_HOOKS  // the chunked path at test sizes (libmjhmc_hip_test.so only)
  if (const char* e = std::getenv("MJHMC_AUTOCOR_STAGING_MB")) {
    const long mb = std::atol(e);
    if (mb > 0) return (size_t)mb << 20;
  }
#endif
  return (size_t)1 << 30;
}

// source of one chunk: fills in[chunk][M] on the stream
struct ChunkSource {
  const RingView* ring = nullptr;
  const double* host = nullptr;
};

// test build: MJHMC_AUTOCOR_TRANSFORM=1 sends short rings through the transform pipeline too (the A/B partner of run_direct)
bool force_transform() {
#ifdef MJHMC_TEST_HOOKS
  return std::getenv("MJHMC_AUTOCOR_TRANSFORM") != nullptr;
#else
  return false;
#endif
}

template <int TT>
int run_direct_t(hipStream_t st, const RingView& r, int64_t n_series, int T, double* lin, std::string& err) {
  int device = 0, cus = 0;
  ACHK(hipGetDevice(&device));
  ACHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
  const int n_blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n_series + 255) / 256, (int64_t)std::max(1, cus) * 8));
  DevBuf partial, out;
  ACHK(hipMalloc(&partial.p, (size_t)n_blocks * TT * sizeof(double)));
  ACHK(hipMalloc(&out.p, TT * sizeof(double)));
  const size_t slot_elems = (size_t)r.Npad * r.pitch;
  if (r.dtype == MJHMC_F64)
    hipLaunchKernelGGL((lag_sums_direct<double, TT>), dim3(n_blocks), dim3(256), 0, st, (const double*)r.base, slot_elems, r.pitch, r.D,
                       T, n_series, (double*)partial.p);
  else if (r.dtype == MJHMC_F32)
    hipLaunchKernelGGL((lag_sums_direct<float, TT>), dim3(n_blocks), dim3(256), 0, st, (const float*)r.base, slot_elems, r.pitch, r.D, T,
                       n_series, (double*)partial.p);
  else
    hipLaunchKernelGGL((lag_sums_direct<__bf16, TT>), dim3(n_blocks), dim3(256), 0, st, (const __bf16*)r.base, slot_elems, r.pitch, r.D,
                       T, n_series, (double*)partial.p);
  ACHK(hipGetLastError());
  hipLaunchKernelGGL(lag_sums_finish<TT>, dim3(1), dim3(64), 0, st, (const double*)partial.p, n_blocks, (double*)out.p);
  ACHK(hipGetLastError());
  double h[TT];
  ACHK(hipMemcpyAsync(h, out.p, TT * sizeof(double), hipMemcpyDeviceToHost, st));
  ACHK(hipStreamSynchronize(st));
  for (int k = 0; k < T; ++k) lin[k] = h[k];
  return 0;
}

int run_direct(hipStream_t st, const RingView& r, int64_t n_series, int T, int linear, double* host_out, std::string& err) {
  double lin[32];
  int rc;
  if (T <= 8) rc = run_direct_t<8>(st, r, n_series, T, lin, err);
  else if (T <= 16) rc = run_direct_t<16>(st, r, n_series, T, lin, err);
  else rc = run_direct_t<32>(st, r, n_series, T, lin, err);
  if (rc) return rc;
  if (linear) {
    for (int k = 0; k < T; ++k) host_out[k] = lin[k];
  } else {   // the time index wraps modulo T: the lag-k products that wrap are the lag-(T - k) products
    host_out[0] = lin[0];
    for (int k = 1; k < T; ++k) host_out[k] = lin[k] + lin[T - k];
  }
  return 0;
}

int run(hipStream_t st, const ChunkSource& src, int64_t n_series, int T, int linear, double* host_out,
        std::string& err) {
  if (T < 1 || T > (1 << 20) || n_series < 1) {
    err = "autocorrelation needs 1 <= n_samples <= 2^20 and at least one series";
    return MJHMC_ERR_INVALID;
  }
  if (src.ring && T <= 32 && n_series < ((int64_t)1 << 32) && !force_transform()) return run_direct(st, *src.ring, n_series, T, linear, host_out, err);
  const FftApi* api = fft_api(err);
  if (!api) return MJHMC_ERR_UNSUPPORTED;
  const int M = (linear || T == 1) ? 2 * T : T;  // transform length (a length-1 series is its own zero-padded case)
  const int nF = M / 2 + 1;
  const size_t per_series = (size_t)M * sizeof(double) + (size_t)nF * sizeof(double2);
  int64_t chunk = (int64_t)(staging_budget() / per_series) / 64 * 64;
  chunk = std::max<int64_t>(64, std::min<int64_t>(chunk, (n_series + 63) / 64 * 64));
  chunk = std::min<int64_t>(chunk, (int64_t)1 << 24);

  DevBuf in, spec, power, ac;
  ACHK(hipMalloc(&in.p, (size_t)chunk * M * sizeof(double)));
  ACHK(hipMalloc(&spec.p, (size_t)chunk * nF * sizeof(double2)));
  ACHK(hipMalloc(&power.p, (size_t)nF * sizeof(double2)));
  ACHK(hipMalloc(&ac.p, (size_t)M * sizeof(double)));
  ACHK(hipMemsetAsync(power.p, 0, (size_t)nF * sizeof(double2), st));

  static thread_local PlanPair plans;
  int device = 0;
  ACHK(hipGetDevice(&device));
  if (!(plans.fwd_live && plans.inv_live && plans.device == device && plans.M == M && plans.chunk == chunk)) {
    plans.release();
    plans.api = api;
    int len[1] = {M};
    FCHK(api->planMany(&plans.fwd, 1, len, nullptr, 1, M, nullptr, 1, nF, HIPFFT_D2Z, (int)chunk));
    plans.fwd_live = true;
    FCHK(api->planMany(&plans.inv, 1, len, nullptr, 1, nF, nullptr, 1, M, HIPFFT_Z2D, 1));
    plans.inv_live = true;
    plans.device = device;
    plans.M = M;
    plans.chunk = chunk;
  }
  FCHK(api->setStream(plans.fwd, st));
  FCHK(api->setStream(plans.inv, st));

  for (int64_t s0 = 0; s0 < n_series; s0 += chunk) {
    const int64_t ns = std::min<int64_t>(chunk, n_series - s0);
    if (src.ring) {
      const RingView& r = *src.ring;
      const dim3 grid((unsigned)(chunk / 64), (M + 63) / 64), block(256);
      const size_t slot_elems = (size_t)r.Npad * r.pitch;
      if (r.dtype == MJHMC_F64)
        hipLaunchKernelGGL(gather_series<double>, grid, block, 0, st, (const double*)r.base, slot_elems, r.pitch, r.D, T,
                           M, s0, ns, (double*)in.p);
      else if (r.dtype == MJHMC_F32)
        hipLaunchKernelGGL(gather_series<float>, grid, block, 0, st, (const float*)r.base, slot_elems, r.pitch, r.D, T,
                           M, s0, ns, (double*)in.p);
      else
        hipLaunchKernelGGL(gather_series<__bf16>, grid, block, 0, st, (const __bf16*)r.base, slot_elems, r.pitch, r.D, T,
                           M, s0, ns, (double*)in.p);
      ACHK(hipGetLastError());
    } else {
      if (ns < chunk || M != T) ACHK(hipMemsetAsync(in.p, 0, (size_t)chunk * M * sizeof(double), st));
      ACHK(hipMemcpy2DAsync(in.p, (size_t)M * sizeof(double), src.host + (size_t)s0 * T, (size_t)T * sizeof(double),
                            (size_t)T * sizeof(double), (size_t)ns, hipMemcpyHostToDevice, st));
    }
    FCHK(api->execD2Z(plans.fwd, (hipfftDoubleReal*)in.p, (hipfftDoubleComplex*)spec.p));
    const int64_t rows = (ns + 63) / 64 * 64;  // rows past ns are zero series
    const unsigned split = (unsigned)std::max<int64_t>(1, std::min<int64_t>(rows, 2048 / ((nF + 255) / 256)));
    hipLaunchKernelGGL(power_accumulate, dim3((nF + 255) / 256, split), dim3(256), 0, st, (const double2*)spec.p, nF,
                       rows, (double2*)power.p);
    ACHK(hipGetLastError());
  }
  FCHK(api->execZ2D(plans.inv, (hipfftDoubleComplex*)power.p, (hipfftDoubleReal*)ac.p));
  std::vector<double> h((size_t)M);
  ACHK(hipMemcpyAsync(h.data(), ac.p, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, st));
  ACHK(hipStreamSynchronize(st));
  // the unnormalised inverse returns M * (circular correlation of the length-M series)
  for (int k = 0; k < T; ++k) host_out[k] = h[(size_t)k] / (double)M;
  return 0;
}

}  // namespace

int autocor_from_ring(hipStream_t st, const RingView& r, int T, int linear, double* host_out, std::string& err) {
  ChunkSource src;
  src.ring = &r;
  return run(st, src, r.N * (int64_t)r.D, T, linear, host_out, err);
}

int autocor_from_host(hipStream_t st, const double* samples, int64_t n_series, int T, int linear, double* host_out,
                      std::string& err) {
  ChunkSource src;
  src.host = samples;
  return run(st, src, n_series, T, linear, host_out, err);
}
