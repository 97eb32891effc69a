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

// Box-Muller pair for dims (2*pair, 2*pair+1) of particle pid
__device__ inline void normal_pair(const RngKey& k, uint32_t pid, uint32_t pair, double& z0, double& z1) {
  const u32x4 w = philox4x32_10(pid, k.tick_lo, k.tick_hi, pair, k.k0, k.k1);
  const double u1 = u53(w.w0, w.w1);
  const double u2 = u53(w.w2, w.w3);
  const double r = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(2.0 * u2, &s, &c);  // angle 2*pi*u2; the pi-scaled form needs no Payne-Hanek reduction
  z0 = r * c;
  z1 = r * s;
}

}  // namespace mjhmc
