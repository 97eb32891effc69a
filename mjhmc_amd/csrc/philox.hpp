// Philox4x32-10 counter RNG (Salmon et al., SC'11) keyed by global particle id.
// Replaces the reference's serial process-global MT19937 stream (np.random.randn in
// mjhmc/samplers/hmc_state.py:26,125; np.random.exponential in mjhmc/misc/utils.py:42), which
// cannot be consumed by 10^5..10^6 independent lanes.  Bit-level recipe mirrored by
// oracle/philox.py.
#pragma once
#ifndef __HIPCC_RTC__  // hipRTC pre-includes the device runtime and has no header search path to it
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace mjhmc {

constexpr uint32_t kSlotExpLF = 0x80000000u;  // words 01 -> e_L (e_FL), words 23 -> e_F
constexpr uint32_t kSlotExpR = 0x80000001u;   // words 01 -> e_R, words 23 -> accept uniform (control)
constexpr uint32_t kSlotFlip = 0x80000002u;   // words 01 -> flip uniform (control)

struct u32x4 {
  uint32_t w0, w1, w2, w3;
};

__host__ __device__ inline u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}

// two words -> double in (0, 1]; every step exact
__host__ __device__ inline double u53(uint32_t a, uint32_t b) {
  const uint64_t x = ((uint64_t)(a >> 5) << 26) + (uint64_t)(b >> 6);
  return ((double)x + 1.0) * (1.0 / 9007199254740992.0);
}

struct RngKey {
  uint32_t k0, k1;        // seed
  uint32_t tick_lo, tick_hi;
};

// -log(u) for u in (0, 1), the unit exponential behind draw_from (utils.py:42): fdlibm's e_log.c scheme
// (error < 1 ulp) without its special cases -- u53() never returns 0 or 1 and its smallest value 2^-54 is a
// normal number.  About half the vector instructions and half the dependent-chain length of the library log
// (which carries double-double intermediates).  (Feeding the coefficients from scalar registers through inline
// asm saved more instructions but cost ~30 VGPRs and an occupancy step: measured slower.)
__device__ __forceinline__ double neg_log_unit(double u) {
  __builtin_amdgcn_sched_barrier(0);  // chains kept compact: freely interleaved they were measured slower
  double m = __builtin_amdgcn_frexp_mant(u);  // [0.5, 1)
  int k = __builtin_amdgcn_frexp_exp(u);
  const bool low = m < __longlong_as_double(0x3fe6a09e667f3bcdLL);  // sqrt(1/2)
  m = low ? m + m : m;  // [sqrt(1/2), sqrt(2))
  k = low ? k - 1 : k;
  const double f = m - 1.0;
  const double t = 2.0 + f;
  double rc = __builtin_amdgcn_rcp(t);  // s = f / t: reciprocal, two Newton steps, one residual correction
  rc = __builtin_fma(__builtin_fma(-t, rc, 1.0), rc, rc);
  rc = __builtin_fma(__builtin_fma(-t, rc, 1.0), rc, rc);
  double sq = f * rc;
  sq = __builtin_fma(__builtin_fma(-t, sq, f), rc, sq);
  const double z = sq * sq;
  const double w = z * z;
  double t1 = w * __longlong_as_double(0x3fc39a09d078c69fLL) + __longlong_as_double(0x3fcc71c51d8e78afLL);  // Lg6, Lg4
  t1 = w * __builtin_fma(w, t1, __longlong_as_double(0x3fd999999997fa04LL));                                        // Lg2
  double t2 = w * __longlong_as_double(0x3fc2f112df3e5244LL) + __longlong_as_double(0x3fc7466496cb03deLL);  // Lg7, Lg5
  t2 = __builtin_fma(w, t2, __longlong_as_double(0x3fd2492494229359LL));                                            // Lg3
  t2 = z * __builtin_fma(w, t2, __longlong_as_double(0x3fe5555555555593LL));                                        // Lg1
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)k;
  const double ln2_hi = __longlong_as_double(0x3fe62e42fee00000LL), ln2_lo = __longlong_as_double(0x3dea39ef35793c76LL);
  const double e = ((hfsq - (sq * (hfsq + R) + dk * ln2_lo)) - f) - dk * ln2_hi;
  __builtin_amdgcn_sched_barrier(0);  // chains kept compact: freely interleaved they were measured slower
  return e;
}


// sin(pi t), cos(pi t) for t in [0, 2]: the quadrant reduction is exact (t is a multiple of 2^-52), then fdlibm's
// k_sin.c / k_cos.c polynomials on |x| <= pi/4 (error ~1 ulp; tools/check_device_math.hip).  About a third of
// the library sincospi's vector instructions: no argument classes to tell apart.
__device__ __forceinline__ void sincospi_unit(double t, double& s, double& c) {
  const double n = __builtin_rint(2.0 * t);                                   // 0 .. 4
  const double r = __builtin_fma(n, -0.5, t);                                 // exact, [-1/4, 1/4]
  const double x = __builtin_fma(r, 1.2246467991473532e-16, r * 3.141592653589793);   // pi = hi + lo
  const double z = x * x;
  double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
  ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
  ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
  const double s0 = __builtin_fma(z * x, __builtin_fma(z, ps, -1.66666666666666324348e-01), x);
  double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
  pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
  pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
  pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
  // k_cos.c's split 1 - x^2/2 = (1 - qx) - (x^2/2 - qx), qx ~ |x| / 4 with an all-zero low word: 1 - qx is exact
  const double ax = __builtin_fabs(x);
  const double qx = ax < 0.3 ? 0.0 : __hiloint2double(__double2hiint(ax) - 0x00200000, 0);
  const double c0 = (1.0 - qx) - ((0.5 * z - qx) - z * (z * pc));
  const int q = (int)n;
  const double ss = (q & 1) ? c0 : s0, cc = (q & 1) ? s0 : c0;
  s = (q & 2) ? -ss : ss;                  // q = 0: (s0, c0)  1: (c0, -s0)  2: (-s0, -c0)  3: (-c0, s0)  4: as 0
  c = ((q + 1) & 2) ? -cc : cc;
}

// Box-Muller pair for dims (2*pair, 2*pair+1) of particle pid: sqrt(-2 log u1) * (cos, sin)(2 pi u2), as oracle/philox.py
__device__ inline void normal_pair(const RngKey& k, uint32_t pid, uint32_t pair, double& z0, double& z1) {
  const u32x4 w = philox4x32_10(pid, k.tick_lo, k.tick_hi, pair, k.k0, k.k1);
  const double u1 = u53(w.w0, w.w1);
  const double u2 = u53(w.w2, w.w3);
  const double r = sqrt(2.0 * neg_log_unit(u1));
  double s, c;
  sincospi_unit(2.0 * u2, s, c);
  z0 = r * c;
  z1 = r * s;
}

}  // namespace mjhmc
