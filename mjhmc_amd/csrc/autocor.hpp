// Autocorrelation along the time axis of the sample ring: the step that follows the hot path in every
// experiment of the reference (mjhmc/misc/autocor.py:37-49 fft_autocor, :177-211 slow_autocorrelation).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

// time-major snapshots of the particle-major state matrix: element (t, particle p, dim d) at
// base[(t * Npad + p) * pitch + d], dtype one of MJHMC_F64 / MJHMC_F32 / MJHMC_BF16
struct RingView {
  const void* base;
  int dtype;
  int64_t Npad, N;
  int D, pitch;
};

// out[k] = sum over series (d < D, p < N) of sum_t x[t] * x[t + k]; the time index wraps modulo T when
// linear == 0 (the cross-correlation theorem on length-T transforms, as fft_autocor), and stops at T when
// linear != 0 (zero-padded transforms: the lag products of slow_autocorrelation).  T doubles to the host.
int autocor_from_ring(hipStream_t st, const RingView& r, int T, int linear, double* host_out, std::string& err);
// same for a host array of n_series contiguous length-T series (the reference's [n_dims, n_batch, n_samples])
int autocor_from_host(hipStream_t st, const double* samples, int64_t n_series, int T, int linear, double* host_out,
                      std::string& err);
