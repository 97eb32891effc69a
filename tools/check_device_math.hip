// Accuracy check of the hand-written transcendental kernels of the jump kernel against host libm.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -ffp-contract=off -DMJHMC_JUMP_WAVES=1 -I mjhmc_amd/csrc tools/check_device_math.hip -o /tmp/cdm && /tmp/cdm
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "elementwise.hpp"

__global__ void eval(const double* dH, const double* u, double* rate, double* nl, double* ex, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    rate[i] = mjhmc::jump_rate(dH[i]);
    nl[i] = mjhmc::neg_log_unit(u[i]);
    ex[i] = mjhmc::exp_any(dH[i]);
  }
}

__global__ void eval_trig(const double* t, double* s, double* c, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) mjhmc::sincospi_unit(t[i], s[i], c[i]);
}

static double ulps(double got, double want) {
  if (std::isnan(got) && std::isnan(want)) return 0;
  if (got == want) return 0;
  if (!std::isfinite(got) || !std::isfinite(want)) return 1e300;
  long long a, b;
  std::memcpy(&a, &got, 8);
  std::memcpy(&b, &want, 8);
  return (double)std::llabs(a - b);
}

int main() {
  std::vector<double> dH, u;
  for (double x = -760; x <= 720; x += 0.0137) dH.push_back(x);
  for (double x : {-745.2, -745.13, -745.0, -708.4, -708.0, -707.99, 0.0, 1e-300, -1e-300, 708.99, 709.0, 709.78, 709.79, 710.0,
                   1e308, -1e308, (double)INFINITY, -(double)INFINITY, (double)NAN})
    dH.push_back(x);
  unsigned long long st = 88172645463325252ULL;
  while (u.size() < dH.size()) {
    st ^= st << 13;
    st ^= st >> 7;
    st ^= st << 17;
    const double v = ((double)(st >> 11) + 0.5) * 0x1p-53;
    u.push_back(u.size() % 5 == 0 ? v * 0x1p-30 : (u.size() % 7 == 0 ? 1.0 - v * 0x1p-20 : v));
  }
  u[0] = 0x1p-54;
  u[1] = 1.0 - 0x1p-54;
  const int n = (int)dH.size();
  double *d_dH, *d_u, *d_r, *d_l, *d_e;
  hipMalloc(&d_e, n * 8);
  hipMalloc(&d_dH, n * 8);
  hipMalloc(&d_u, n * 8);
  hipMalloc(&d_r, n * 8);
  hipMalloc(&d_l, n * 8);
  hipMemcpy(d_dH, dH.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(d_u, u.data(), n * 8, hipMemcpyHostToDevice);
  eval<<<(n + 255) / 256, 256>>>(d_dH, d_u, d_r, d_l, d_e, n);
  std::vector<double> r(n), l(n), ex(n);
  hipMemcpy(ex.data(), d_e, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r.data(), d_r, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(l.data(), d_l, n * 8, hipMemcpyDeviceToHost);
  double worst_r = 0, worst_l = 0, worst_special = 0, worst_e = 0;
  for (int i = 0; i < n; ++i) {
    const double want_r = std::sqrt(std::exp(dH[i]));
    const double e = ulps(r[i], want_r);
    const bool fast = dH[i] > -708.0 && dH[i] < 709.0;
    if (fast) worst_r = std::fmax(worst_r, e);
    else worst_special = std::fmax(worst_special, e);
    worst_l = std::fmax(worst_l, ulps(l[i], -std::log(u[i])));
    worst_e = std::fmax(worst_e, ulps(ex[i], std::exp(dH[i])));
  }
  std::printf("n=%d  jump_rate: max %.0f ulp on the fast range, %.0f ulp on the literal ranges;  neg_log_unit: max %.0f ulp;  "
              "exp_any: max %.0f ulp (subnormal, inf, 0 and NaN results included)\n", n, worst_r, worst_special, worst_l, worst_e);
  // sincospi_unit on [0, 2] (the Box-Muller angle 2 u2): against long double sin / cos of pi t, absolute error in units
  // of 2^-53 (the values are <= 1 in magnitude; that is one ulp of [1/2, 1))
  std::vector<double> tt;
  for (int k = 0; k <= 400000; ++k) tt.push_back(2.0 * k / 400000.0);
  for (double v : u) tt.push_back(2.0 * v);
  const int m = (int)tt.size();
  double *d_t, *d_s, *d_c;
  hipMalloc(&d_t, m * 8);
  hipMalloc(&d_s, m * 8);
  hipMalloc(&d_c, m * 8);
  hipMemcpy(d_t, tt.data(), m * 8, hipMemcpyHostToDevice);
  eval_trig<<<(m + 255) / 256, 256>>>(d_t, d_s, d_c, m);
  std::vector<double> ss(m), cc(m);
  hipMemcpy(ss.data(), d_s, m * 8, hipMemcpyDeviceToHost);
  hipMemcpy(cc.data(), d_c, m * 8, hipMemcpyDeviceToHost);
  double worst_t = 0;
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int i = 0; i < m; ++i) {
    const long double ws = sinl(pi * (long double)tt[i]), wc = cosl(pi * (long double)tt[i]);
    const double es = (double)fabsl((long double)ss[i] - ws) * 0x1p53;   // in units of 2^-53 = one ulp of [1/2, 1)
    const double ec = (double)fabsl((long double)cc[i] - wc) * 0x1p53;
    worst_t = std::fmax(worst_t, std::fmax(es, ec));
  }
  std::printf("sincospi_unit: max error %.2f x 2^-53 (one ulp of [1/2, 1)) over %d angles\n", worst_t, m);
  // the literal ranges run the device library's exp and sqrt (subnormal results: a few ulp from glibc's)
  return (worst_r <= 2 && worst_special <= 16 && worst_l <= 2 && worst_e <= 2 && worst_t <= 1.5) ? 0 : 1;
}
