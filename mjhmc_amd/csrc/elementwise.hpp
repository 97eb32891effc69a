// Fused Markov-jump kernel for energies whose gradient is elementwise (optionally coupled through
// a few per-particle scalars).  One launch = one MarkovJumpHMC.sampling_iteration attempt
// (mjhmc/samplers/markov_jump_hmc.py:355-415) for every particle:
//
//   load (X, V) once -> inverse-L proposal (only if the FLF cache is cold; only its H is kept,
//   because the reference reads the FLF state only through H(), :367) -> forward L proposal
//   (mjhmc/samplers/hmc_state.py:86-100) -> transition rates (:341-347) -> waiting-time draws
//   (mjhmc/misc/utils.py:31-49) -> first-minimum (utils.py:15-28) -> write the chosen
//   successor state once.
//
// Data layout in HBM: particle-major.  Particle p owns `pitch` consecutive elements, so the 64/G
// particles a wavefront works on are ONE contiguous span and every global access is a full
// 16 B/lane, 1 KiB/wave-instruction transaction.  G (power of two, <= 64) lanes share a particle;
// lane j holds chunks j, j+G, j+2G, ... (16-byte chunks) of X and V in registers for the whole
// iteration -- both trajectories run out of registers, reductions are xor-butterflies inside the
// G-lane group, and for G == 64 (ndims >= 256) all per-particle control flow is wave-uniform.
//
// Floating-point contraction is OFF for this translation unit (see Makefile): the leapfrog
// update is evaluated as the reference writes it (mul, then add), which makes X and V
// bit-identical to the NumPy path for linear forces.
#pragma once
#ifndef __HIPCC_RTC__  // hipRTC (user-expression energies, user_expr.hip) compiles the device half of this header only
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#endif
#include <stdint.h>

#include "philox.hpp"
#include "timing_variants.hpp"

#ifndef MJHMC_NT
#define MJHMC_NT 1  // streaming (nontemporal) row loads/stores: +1-2 % on C2, state rows are touched once per launch
#endif

namespace mjhmc {

constexpr int kModeMJHMC = 0;    // MarkovJumpHMC.sampling_iteration      (markov_jump_hmc.py:355-415)
constexpr int kModeControl = 1;  // HMCBase / HMC / ControlHMC            (markov_jump_hmc.py:116-148)
constexpr int kModeCT = 2;       // ContinuousTimeHMC.sampling_iteration  (markov_jump_hmc.py:251-290)

struct Control {
  int failed;       // some attempt hit a non-finite rate
  int failed_iter;  // index (within the current mjhmc_iterate call) of that attempt
  int inv_iter;     // fused launches: 0x7fffffff - (first failed attempt), combined with atomicMax (0 = none)
};

// most sampling iterations one fused launch runs per particle (per-iteration tallies live in LDS)
// Dense energies (tile kernels): the particles whose inverse-L cache will be cold in the NEXT iteration are known when the
// jump kernel has decided -- every move but L clears the cache (markov_jump_hmc.py:398-412) -- so the lanes that decided
// (one per particle) append them to the next iteration's compacted list right there: one atomic per tile.  Called under
// divergent control flow; `cold` lanes must be active.
__device__ __forceinline__ void append_cold(int* list, int* count, bool cold, int64_t p) {
  const unsigned long long m = __ballot(cold);
  if (m == 0ull) return;
  const int lane = threadIdx.x & 63;
  const int first = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane == first) base = atomicAdd(count, (int)__popcll(m));
  base = __shfl(base, first);
  if (cold) list[base + (int)__popcll(m & ((1ull << lane) - 1ull))] = (int)p;
}

constexpr int kMaxFuse = 64;

template <typename T>
struct VecOf;
template <>
struct VecOf<double> {
  using type = double2;
  static constexpr int n = 2;
};
template <>
struct VecOf<float> {
  using type = float4;
  static constexpr int n = 4;
};

// which dims a lane owns
struct LaneMap {
  int j;      // lane within the particle group
  int G;      // lanes per particle
  int D;      // true ndims
  int CH;     // 16-byte chunks per particle row (pitch / VEC)
  int lane0;  // wave lane index of the group's lane 0
  int wpp;    // lane mapping known at compile time: 1 = the wavefront is one particle (G == 64), 3 = a quad per particle
              // (G == 4), 6 = half a wavefront per particle (G == 32); 0 = G is a run-time value
};

template <typename T, int E>
__device__ __forceinline__ int dim_of(const LaneMap& m, int e) {
  constexpr int VEC = VecOf<T>::n;
  return ((e / VEC) * m.G + m.j) * VEC + (e % VEC);
}

// ---- cross-lane primitives -------------------------------------------------------------------
// Sums inside a G-lane group run on the VALU cross-lane network (DPP quad_perm / row mirrors inside
// a 16-lane row, gfx950's v_permlane16_swap / v_permlane32_swap across rows): no LDS crossbar
// round trips (ds_bpermute), which cost ~100+ cycles of latency per butterfly step.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  // bound_ctrl = true with full row / bank masks: every lane is written, so no zero-initialised destination
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ double swap16_sum(double v) {
  const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ float swap16_sum(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ double swap32_sum(double v) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ float swap32_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
  return __int_as_float(r[0]) + __int_as_float(r[1]);
}

// sum over the lanes of a group (G = 2^k <= 64, groups aligned); every lane gets the same bits.
// (A branch-free form that always runs all six levels and selects zeros was measured: slower, the
// permlane swaps cost more than the wave-uniform branches they replace.)
template <typename T>
__device__ __forceinline__ T group_sum(T v, int G) {
  if (G > 32) v = swap32_sum(v);       // other half of the wave FIRST (the order group_sum_pair reproduces)
  if (G > 1) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]  : lane ^ 1
  if (G > 2) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]  : lane ^ 2
  if (G > 4) v += dpp_mov<0x141>(v);   // row_half_mirror      : other quad of the 8
  if (G > 8) v += dpp_mov<0x140>(v);   // row_mirror           : other 8 of the row
  if (G > 16) v = swap16_sum(v);       // other row of the pair
  return v;
}

// v_permlane32_swap on every dword of the pair: afterwards a = [a.lower half | b.lower half], b = [a.upper | b.upper]
__device__ __forceinline__ void swap32_pair(double& a, double& b) {
  const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
  a = __hiloint2double(hi[0], lo[0]);
  b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void swap32_pair(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
  a = __int_as_float(r[0]);
  b = __int_as_float(r[1]);
}

// group_sum of TWO values at once.  With a wavefront per particle the two reductions share their ladder: one half-wave
// swap leaves a's pair sums in lanes 0..31 and b's in lanes 32..63, five levels reduce both inside their halves, one
// more swap hands each total to the other half -- 24 vector instructions in float64 instead of 44, and bit for bit
// what two group_sum calls return (same pairing at every level).
template <typename T>
__device__ __forceinline__ void group_sum_pair(T a, T b, int G, T& ta, T& tb) {
  if (G == 64) {
    swap32_pair(a, b);  // a = [a_l | b_l], b = [a_{l+32} | b_{l+32}]
    T c = a + b;
    c += dpp_mov<0xB1>(c);
    c += dpp_mov<0x4E>(c);
    c += dpp_mov<0x141>(c);
    c += dpp_mov<0x140>(c);
    c = swap16_sum(c);
    ta = c;
    tb = c;
    swap32_pair(ta, tb);  // ta = [total a | total a], tb = [total b | total b]
  } else {
    ta = group_sum(a, G);
    tb = group_sum(b, G);
  }
}

// value held by lane `r` of the caller's group
__device__ __forceinline__ double readlane_v(double v, int r) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), r), __builtin_amdgcn_readlane(__double2loint(v), r));
}
__device__ __forceinline__ float readlane_v(float v, int r) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), r));
}

template <typename T>
__device__ __forceinline__ T group_lane(T v, const LaneMap& m, int r) {
  if (m.wpp == 1) return readlane_v(v, r);  // v_readlane: no LDS-crossbar round trip
  if (m.wpp == 6) {                         // half a wavefront per particle: both halves' lane r, then this lane's half
    const T lo = readlane_v(v, r), hi = readlane_v(v, 32 + r);
    return m.lane0 ? hi : lo;
  }
  if (m.wpp == 3) {                         // lane r of the quad: one DPP quad_perm broadcast
    switch (r) {
      case 0: return dpp_mov<0x00>(v);
      case 1: return dpp_mov<0x55>(v);
      case 2: return dpp_mov<0xAA>(v);
      default: return dpp_mov<0xFF>(v);
    }
  }
  return __shfl(v, m.lane0 + r);
}

// value of the group's lane 0.  Groups of <= 4 lanes sit inside one quad: a DPP quad_perm broadcast
// (VALU, no LDS crossbar round trip); wider groups go through ds_bpermute.
template <typename T>
__device__ __forceinline__ T group_bcast0(T v, const LaneMap& m) {
  if (m.G == 1) return v;
  if (m.G == 2) return dpp_mov<0xA0>(v);  // quad_perm [0,0,2,2]
  if (m.G == 4) return dpp_mov<0x00>(v);  // quad_perm [0,0,0,0]
  return __shfl(v, m.lane0);
}

// ------------------------------------------------------------------------------------------
// energy functors.  Protocol (E = elements per lane):
//   Local<E> local<E>(m)              per-lane constants, loaded once per kernel
//   Ctx      prep(x, m)               per-evaluation context (may reduce over the group)
//   T        grad(xe, e, d, ctx, lc)  dE/dx_d for the lane's element e (dim d)
//   T        energy(x, m, lc)         E(x), reduced over the group, valid in every lane
// Padded elements (d >= D) hold x = 0 and must produce grad = 0; energy() masks them.
// ------------------------------------------------------------------------------------------

// defined with the other scalar kernels below
__device__ __forceinline__ double exp_any(double h);
__device__ __forceinline__ float exp_any(float h);
__device__ __forceinline__ double exp_neg(double x);
__device__ __forceinline__ float exp_neg(float x);
// exp_neg()'s fifteen constants held in VECTOR registers by the caller for the length of a kernel (exp_neg_k: the same
// operations on the same constants, the same bits).  The relay kernel's step loops re-materialised them in scalar registers
// at every leapfrog step -- 30 s_mov_b32 per step: its 106 scalar registers are taken (70 spilled) and the loop's constants
// lost -- where it has vector registers to spare.
struct ExpNegK {
  double k[15];
};
struct NoExpK {};
__device__ __forceinline__ ExpNegK exp_neg_pinned();
__device__ __forceinline__ double exp_neg_k(double x, const ExpNegK& K);
__device__ __forceinline__ double exp_neg_k(double x, const NoExpK&) { return exp_neg(x); }

struct NoCtx {};
template <int E>
struct NoLocal {};

// TestGaussian: E = sum(x^2) / (2 sigma^2), dE/dx = x / sigma^2 (distributions.py:356-362)
template <typename T>
struct IsoGaussF {
  // fused launches (several iterations per launch) exist for this energy; WHEN they are used is the host's decision
  // (api.hip, iterate_t): always for the Gaussian forces, whose single iteration is HBM-bound (measured 1.03-2.7x at
  // every row size), and for the vector-pipe-bound ones only while the batch is small (launch- and latency-bound:
  // funnel 18.7 -> 12.7 us per iteration at 1000 particles; 0.93-0.99x at 10^6, where the compacted passes win)
  static constexpr bool kFuse = true;
  // the force is one multiplication by a constant: the half-kick factor is folded into it,
  // c * (x * inv_s2) -> x * (c * inv_s2)  (identical bits when sigma is a power of two, e.g. the benchmark's 1)
  static constexpr bool kLinearIso = true;
  T inv_s2;      // 1 / sigma^2
  T inv_two_s2;  // 1 / (2 sigma^2)
  using Ctx = NoCtx;
  template <int E>
  using Local = NoLocal<E>;
  template <int E>
  __device__ __forceinline__ Local<E> local(const LaneMap&) const {
    return {};
  }
  template <int E>
  __device__ __forceinline__ Ctx prep(const T (&)[E], const LaneMap&) const {
    return {};
  }
  template <int E>
  __device__ __forceinline__ T grad(T xe, int, int, const Ctx&, const Local<E>&) const {
    return xe * inv_s2;
  }
  // energy = energy_scale(group_sum(energy_lane)): lets the kernels reduce it together with the kinetic energy
  static constexpr bool kLaneEnergy = true;
  template <int E>
  __device__ __forceinline__ T energy_lane(const T (&x)[E], const LaneMap&, const Local<E>&) const {
    T s = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) s = __builtin_fma(x[e], x[e], s);  // padded x are 0; sums are compared at 1e-10, not bitwise
    return s;
  }
  __device__ __forceinline__ T energy_scale(T total) const { return total * inv_two_s2; }
  template <int E>
  __device__ __forceinline__ T energy(const T (&x)[E], const LaneMap& m, const Local<E>& lc) const {
    return energy_scale(group_sum(energy_lane<E>(x, m, lc), m.G));
  }
};

// Gaussian with diagonal J: E = sum x (J x) / 2, dE/dx = Jx/2 + J^T x/2 == j_d x_d exactly
// (distributions.py:268-273)
template <typename T>
struct DiagGaussF {
  static constexpr bool kLinearIso = false;
  static constexpr bool kFuse = true;
  const T* jdiag;  // zero padded to kParamPad elements
  using Ctx = NoCtx;
  template <int E>
  struct Local {
    T j[E];
  };
  template <int E>
  __device__ __forceinline__ Local<E> local(const LaneMap& m) const {
    Local<E> lc;
#pragma unroll
    for (int e = 0; e < E; ++e) lc.j[e] = jdiag[dim_of<T, E>(m, e)];
    return lc;
  }
  template <int E>
  __device__ __forceinline__ Ctx prep(const T (&)[E], const LaneMap&) const {
    return {};
  }
  template <int E>
  __device__ __forceinline__ T grad(T xe, int e, int, const Ctx&, const Local<E>& lc) const {
    return lc.j[e] * xe;
  }
  static constexpr bool kLaneEnergy = true;
  template <int E>
  __device__ __forceinline__ T energy_lane(const T (&x)[E], const LaneMap&, const Local<E>& lc) const {
    T s = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) s = __builtin_fma(x[e], lc.j[e] * x[e], s);
    return s;
  }
  __device__ __forceinline__ T energy_scale(T total) const { return total / T(2); }
  template <int E>
  __device__ __forceinline__ T energy(const T (&x)[E], const LaneMap& m, const Local<E>& lc) const {
    return energy_scale(group_sum(energy_lane<E>(x, m, lc), m.G));
  }
};

// RoughWell: E = sum x^2/(2 s1^2) + cos(2 pi x / s2); dE/dx = x/s1^2 - sin(2 pi x/s2) 2 pi/s2
// (distributions.py:295-304), operation order as written there.
template <typename T>
struct RoughWellF {
  static constexpr bool kLinearIso = false;
  static constexpr bool kFuse = true;
  T s1sq;       // scale1^2
  T two_s1sq;   // 2 scale1^2
  T s2;
  using Ctx = NoCtx;
  template <int E>
  using Local = NoLocal<E>;
  template <int E>
  __device__ __forceinline__ Local<E> local(const LaneMap&) const {
    return {};
  }
  template <int E>
  __device__ __forceinline__ Ctx prep(const T (&)[E], const LaneMap&) const {
    return {};
  }
  template <int E>
  __device__ __forceinline__ T grad(T xe, int, int, const Ctx&, const Local<E>&) const {
    const T pi = T(3.141592653589793);
    const T sn = sin(xe * T(2) * pi / s2);
    return xe / s1sq + -sn * T(2) * pi / s2;
  }
  template <int E>
  __device__ __forceinline__ T energy(const T (&x)[E], const LaneMap& m, const Local<E>&) const {
    const T pi = T(3.141592653589793);
    T s = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const T t = (x[e] * x[e]) / two_s1sq + cos(x[e] * T(2) * pi / s2);
      s += (dim_of<T, E>(m, e) < m.D) ? t : T(0);
    }
    return group_sum(s, m.G);
  }
};

// MultimodalGaussian as coded (distributions.py:323-335): separation vector = (2*sep, 0, ..., 0).
template <typename T>
struct MMGaussF {
  static constexpr bool kLinearIso = false;
  static constexpr bool kFuse = true;
  T sep0;  // 2 * separation
  struct Ctx {
    T common;  // exp(sum 4 S X) = exp(4 sep0 x_0)
  };
  template <int E>
  using Local = NoLocal<E>;
  template <int E>
  __device__ __forceinline__ Local<E> local(const LaneMap&) const {
    return {};
  }
  template <int E>
  __device__ __forceinline__ Ctx prep(const T (&x)[E], const LaneMap& m) const {
    const T x0 = group_bcast0(x[0], m);
    return Ctx{(T)exp(T(4) * sep0 * x0)};
  }
  template <int E>
  __device__ __forceinline__ T grad(T xe, int, int d, const Ctx& c, const Local<E>&) const {
    const T s = (d == 0) ? sep0 : T(0);
    return (T(2) * ((xe - s) * c.common + s + xe)) / (c.common + T(1));
  }
  template <int E>
  __device__ __forceinline__ T energy(const T (&x)[E], const LaneMap& m, const Local<E>&) const {
    T a = 0, b = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const T s = (dim_of<T, E>(m, e) == 0) ? sep0 : T(0);
      a += (x[e] + s) * (x[e] + s);
      b += (x[e] - s) * (x[e] - s);
    }
    a = group_sum(a, m.G);
    b = group_sum(b, m.G);
    return -(T)log(exp(-a) + exp(-b));
  }
  // the row form (round 6; the protocol of the funnels below): the whole particle in one lane -- exp(4 sep x_0), once per
  // particle and leapfrog step, is paid once per particle instead of once per lane of its group -- and, for the relay of
  // mjhmc_fused_rows_relay_kernel, half a row per lane of a pair.  The same operations in the same order as prep / grad /
  // energy: the same bits (tests/test_gpu_fused.py, tools/fuzz_rows.py).
  static constexpr bool kRowForm = true;
  // (a float64 division per coordinate and step: the relay's saved trajectory outweighs its hand-overs from ~4 steps up --
  // 32 x 10^6, L = 5: 0.288 ms in the relay kernel, 0.308 in the group form; the funnels' break-even is 12 steps)
  static constexpr int kRelayMinL = 4;
  template <int E>
  __device__ __forceinline__ T kick(T ck, T xe, T ve, int e, int d, const Ctx& c) const {
    return __builtin_fma(ck, grad<E>(xe, e, d, c, Local<E>{}), ve);   // kick_fma's form for an energy without a scaled kick
  }
  template <int E, int G, class XK = NoExpK>
  __device__ __forceinline__ Ctx prep_rows(const T (&x)[G][E], const XK& = XK{}) const {
    return Ctx{(T)exp(T(4) * sep0 * x[0][0])};
  }
  template <int E, int G, class XK = NoExpK>
  __device__ __forceinline__ Ctx prep_rows_pair(const T (&xh)[G / 2][E], int, const XK& = XK{}) const {
    return Ctx{(T)exp(T(4) * sep0 * dpp_mov<0xA0>(xh[0][0]))};   // quad_perm [0, 0, 2, 2]: the even lane's x[0][0]
  }
  template <int E, int G>
  __device__ __forceinline__ T energy_rows(const T (&x)[G][E]) const {
    T pa[G], pb[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      T a = 0, b = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const T s = (((e / 2) * G + j) * 2 + (e % 2) == 0) ? sep0 : T(0);   // row_dim<G>(j, e) == 0
        a += (x[j][e] + s) * (x[j][e] + s);
        b += (x[j][e] - s) * (x[j][e] - s);
      }
      pa[j] = a;
      pb[j] = b;
    }
    T a, b;
    if constexpr (G == 1) { a = pa[0]; b = pb[0]; }
    else if constexpr (G == 2) { a = pa[0] + pa[1]; b = pb[0] + pb[1]; }
    else { a = (pa[0] + pa[1]) + (pa[2] + pa[3]); b = (pb[0] + pb[1]) + (pb[2] + pb[3]); }   // rows_sum
    return -(T)log(exp(-a) + exp(-b));
  }
  template <int E, int G, class XK = NoExpK>
  __device__ __forceinline__ T energy_pair(const T (&xh)[G / 2][E], int h, const XK& = XK{}) const {
    constexpr int GH = G / 2;
    T pa[GH], pb[GH];
#pragma unroll
    for (int jj = 0; jj < GH; ++jj) {
      T a = 0, b = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const T s = (e == 0 && jj == 0 && h == 0) ? sep0 : T(0);   // dim 0 is the even lane's first coordinate
        a += (xh[jj][e] + s) * (xh[jj][e] + s);
        b += (xh[jj][e] - s) * (xh[jj][e] - s);
      }
      pa[jj] = a;
      pb[jj] = b;
    }
    const T ma = (GH == 1) ? pa[0] : pa[0] + pa[GH - 1], mb = (GH == 1) ? pb[0] : pb[0] + pb[GH - 1];
    const T a = ma + dpp_mov<0xB1>(ma), b = mb + dpp_mov<0xB1>(mb);   // the pair's sums: (p0 + p1) + (p2 + p3), as rows_sum
    return -(T)log(exp(-a) + exp(-b));
  }
};

// group_sum()'s pairing on the G per-lane partial sums of a particle held by ONE lane (the row form): the same bits
template <typename T, int G>
__device__ __forceinline__ T rows_sum(const T (&p)[G]) {
  static_assert(G == 1 || G == 2 || G == 4, "groups inside a quad");
  if constexpr (G == 1) return p[0];
  else if constexpr (G == 2) return p[0] + p[1];
  else return (p[0] + p[1]) + (p[2] + p[3]);
}
// prep() of the two funnels on a whole row: lane j's partial sum exactly as prep() forms it, the partials paired as
// group_sum pairs them, exp(-x0) once per particle (prep() spends the instructions in all G lanes of the group)
template <class Ctx, typename T, int E, int G, class XK = NoExpK>
__device__ __forceinline__ Ctx funnel_prep_rows(const T (&x)[G][E], const XK& xk = XK{}) {
  T part[G];
#pragma unroll
  for (int j = 0; j < G; ++j) {
    T s = (j == 0) ? T(0) : x[j][0] * x[j][0];
#pragma unroll
    for (int e = 1; e < E; ++e) s = __builtin_fma(x[j][e], x[j][e], s);
    part[j] = s;
  }
  Ctx c;
  c.S = rows_sum<T, G>(part);
  c.x0 = x[0][0];
  if constexpr (sizeof(T) == 8) c.ex = exp_neg_k(x[0][0], xk);
  else c.ex = exp_neg(x[0][0]);
  return c;
}

// ... and on HALF a row: the two lanes of a pair hold x[j] for j in [h G/2, (h + 1) G/2), h = 0 (even lane), 1 (odd lane)
// -- the relay of mjhmc_fused_rows_relay_kernel integrates a pooled particle on two lanes.  The partial sums are formed as
// funnel_prep_rows forms them; a pair's sums meet as (p0 + p1) + (p2 + p3) in the even lane and (p2 + p3) + (p0 + p1) in
// the odd one (G = 2: p0 + p1 / p1 + p0) -- the same bits, a floating-point sum does not depend on the order of its TWO terms.
template <class Ctx, typename T, int E, int G, class XK = NoExpK>
__device__ __forceinline__ Ctx funnel_prep_pair(const T (&xh)[G / 2][E], int h, const XK& xk = XK{}) {
  constexpr int GH = G / 2;
  static_assert(G == 2 || G == 4, "a row in two halves");
  T part[GH];
#pragma unroll
  for (int jj = 0; jj < GH; ++jj) {
    T s = xh[jj][0] * xh[jj][0];
    if (jj == 0) s = (h == 0) ? T(0) : s;   // (dim 0 is not part of the sum)
#pragma unroll
    for (int e = 1; e < E; ++e) s = __builtin_fma(xh[jj][e], xh[jj][e], s);
    part[jj] = s;
  }
  const T mine = (GH == 1) ? part[0] : part[0] + part[GH - 1];
  Ctx c;
  c.S = mine + dpp_mov<0xB1>(mine);          // quad_perm [1, 0, 3, 2]: the pair's other lane
  c.x0 = dpp_mov<0xA0>(xh[0][0]);            // quad_perm [0, 0, 2, 2]: the even lane's x[0][0]
  if constexpr (sizeof(T) == 8) c.ex = exp_neg_k(c.x0, xk);
  else c.ex = exp_neg(c.x0);
  return c;
}

// Neal's funnel, x0 ~ N(0, s^2), x_k ~ N(0, e^{x0}) (tf_distributions.py:143-147):
// E = x0^2/(2 s^2) + e^{-x0} sum_k x_k^2 / 2 + (D-1) x0 / 2
template <typename T>
struct FunnelNealF {
  static constexpr bool kLinearIso = false;
  static constexpr bool kFuse = true;
  static constexpr int kWavesRow64 = 3;  // JumpWaves
  T inv_s2;      // 1/scale^2
  T half_dm1;    // (D-1)/2
  struct Ctx {
    T x0, ex, S;  // x_0, exp(-x_0), sum_{k>=1} x_k^2
  };
  template <int E>
  using Local = NoLocal<E>;
  template <int E>
  __device__ __forceinline__ Local<E> local(const LaneMap&) const {
    return {};
  }
  template <int E>
  __device__ __forceinline__ Ctx prep(const T (&x)[E], const LaneMap& m) const {
    T s = (m.j == 0) ? T(0) : x[0] * x[0];  // only (lane 0, element 0) is dim 0
#pragma unroll
    for (int e = 1; e < E; ++e) s = __builtin_fma(x[e], x[e], s);
    Ctx c;
    c.S = group_sum(s, m.G);
    c.x0 = group_bcast0(x[0], m);
    // exp(-x0) in the lane that holds x0, then to its group: the other lanes idle through the ~25 instructions instead of
    // repeating them (same value; C4 runs 8 % higher clocks for it at 2 % less power -- and only 0.4 % faster: the kernel
    // is not bound by its clock)
    T ex0 = T(0);
    if (m.j == 0) ex0 = exp_neg(x[0]);
    c.ex = group_bcast0(ex0, m);
    return c;
  }
  template <int E>
  __device__ __forceinline__ T grad(T xe, int e, int d, const Ctx& c, const Local<E>&) const {
    return (e == 0 && d == 0) ? (c.x0 * inv_s2 - T(0.5) * c.ex * c.S + half_dm1) : xe * c.ex;
  }
  // v += ck * dE/dx as ONE multiply-add per coordinate: the force of x_k is x_k * exp(-x0), and ck * exp(-x0) is shared
  // by the whole particle (counter-RNG trajectories; the replay kernels keep c * (x * ex), the reference's rounding)
  static constexpr bool kScaledKick = true;
  template <int E>
  __device__ __forceinline__ T kick(T ck, T xe, T ve, int e, int d, const Ctx& c) const {
    return (e == 0 && d == 0) ? __builtin_fma(ck, c.x0 * inv_s2 - T(0.5) * c.ex * c.S + half_dm1, ve)
                              : __builtin_fma(xe, ck * c.ex, ve);
  }
  __device__ __forceinline__ T energy_of(const Ctx& c) const {
    return c.x0 * c.x0 * (T(0.5) * inv_s2) + T(0.5) * c.ex * c.S + half_dm1 * c.x0;
  }
  template <int E>
  __device__ __forceinline__ T energy(const T (&x)[E], const LaneMap& m, const Local<E>&) const {
    return energy_of(prep(x, m));
  }
  // the row form (mjhmc_traj_rows_kernel): the whole particle in one lane, x[j] = what lane j of its group would hold
  static constexpr bool kRowForm = true;
  template <int E, int G, class XK = NoExpK>
  __device__ __forceinline__ Ctx prep_rows(const T (&x)[G][E], const XK& xk = XK{}) const {
    return funnel_prep_rows<Ctx, T, E, G, XK>(x, xk);
  }
  template <int E, int G, class XK = NoExpK>
  __device__ __forceinline__ Ctx prep_rows_pair(const T (&xh)[G / 2][E], int h, const XK& xk = XK{}) const {
    return funnel_prep_pair<Ctx, T, E, G, XK>(xh, h, xk);
  }
  template <int E, int G>
  __device__ __forceinline__ T energy_rows(const T (&x)[G][E]) const {
    return energy_of(funnel_prep_rows<Ctx, T, E, G>(x));
  }
  template <int E, int G, class XK = NoExpK>
  __device__ __forceinline__ T energy_pair(const T (&xh)[G / 2][E], int h, const XK& xk = XK{}) const {
    return energy_of(funnel_prep_pair<Ctx, T, E, G, XK>(xh, h, xk));
  }
};

// Funnel exactly as coded (tf_distributions.py:157-165): E = -(D-1) x0^2/s^2 - e^{-x0} sum_k x_k^2
template <typename T>
struct FunnelRefF {
  static constexpr bool kLinearIso = false;
  static constexpr bool kFuse = true;
  static constexpr int kWavesRow64 = 3;  // JumpWaves
  T inv_s2;
  T dm1;  // D-1
  struct Ctx {
    T x0, ex, S;
  };
  template <int E>
  using Local = NoLocal<E>;
  template <int E>
  __device__ __forceinline__ Local<E> local(const LaneMap&) const {
    return {};
  }
  template <int E>
  __device__ __forceinline__ Ctx prep(const T (&x)[E], const LaneMap& m) const {
    T s = (m.j == 0) ? T(0) : x[0] * x[0];  // only (lane 0, element 0) is dim 0
#pragma unroll
    for (int e = 1; e < E; ++e) s = __builtin_fma(x[e], x[e], s);
    Ctx c;
    c.S = group_sum(s, m.G);
    c.x0 = group_bcast0(x[0], m);
    // exp(-x0) in the lane that holds x0, then to its group: the other lanes idle through the ~25 instructions instead of
    // repeating them (same value; C4 runs 8 % higher clocks for it at 2 % less power -- and only 0.4 % faster: the kernel
    // is not bound by its clock)
    T ex0 = T(0);
    if (m.j == 0) ex0 = exp_neg(x[0]);
    c.ex = group_bcast0(ex0, m);
    return c;
  }
  template <int E>
  __device__ __forceinline__ T grad(T xe, int e, int d, const Ctx& c, const Local<E>&) const {
    return (e == 0 && d == 0) ? (T(-2) * dm1 * c.x0 * inv_s2 + c.ex * c.S) : T(-2) * xe * c.ex;
  }
  static constexpr bool kScaledKick = true;
  template <int E>
  __device__ __forceinline__ T kick(T ck, T xe, T ve, int e, int d, const Ctx& c) const {
    return (e == 0 && d == 0) ? __builtin_fma(ck, T(-2) * dm1 * c.x0 * inv_s2 + c.ex * c.S, ve)
                              : __builtin_fma(xe, ck * (T(-2) * c.ex), ve);
  }
  __device__ __forceinline__ T energy_of(const Ctx& c) const { return -(dm1 * c.x0 * c.x0 * inv_s2) - c.ex * c.S; }
  template <int E>
  __device__ __forceinline__ T energy(const T (&x)[E], const LaneMap& m, const Local<E>&) const {
    return energy_of(prep(x, m));
  }
  static constexpr bool kRowForm = true;   // as FunnelNealF
  template <int E, int G, class XK = NoExpK>
  __device__ __forceinline__ Ctx prep_rows(const T (&x)[G][E], const XK& xk = XK{}) const {
    return funnel_prep_rows<Ctx, T, E, G, XK>(x, xk);
  }
  template <int E, int G, class XK = NoExpK>
  __device__ __forceinline__ Ctx prep_rows_pair(const T (&xh)[G / 2][E], int h, const XK& xk = XK{}) const {
    return funnel_prep_pair<Ctx, T, E, G, XK>(xh, h, xk);
  }
  template <int E, int G>
  __device__ __forceinline__ T energy_rows(const T (&x)[G][E]) const {
    return energy_of(funnel_prep_rows<Ctx, T, E, G>(x));
  }
  template <int E, int G, class XK = NoExpK>
  __device__ __forceinline__ T energy_pair(const T (&xh)[G / 2][E], int h, const XK& xk = XK{}) const {
    return energy_of(funnel_prep_pair<Ctx, T, E, G, XK>(xh, h, xk));
  }
};

// ------------------------------------------------------------------------------------------
// kernel arguments
// ------------------------------------------------------------------------------------------

// launch_jump_t's A/B flags (JumpArgs::ab): take the generic instance instead of a specialised lane mapping
constexpr int kAbNoBlockDecide = 1, kAbNoWpp = 2, kAbNoQuad = 4, kAbNoRows = 8, kAbNoRelay = 16, kAbForceRelay = 32;

template <typename T>
struct JumpArgs {
  const T* X_in;
  const T* V_in;
  T* X_out;
  T* V_out;
  const T* EX_in;
  const T* EV_in;
  T* EX_out;
  T* EV_out;
  const T* Hflf_in;   // H() of the cached inverse-L state; NaN = cache cold
  T* Hflf_out;
  double* dwell;        // [N]
  double* dwell_ring;   // [Npad] ring slot, or a scratch vector when nothing is recorded
  uint8_t* trans;       // [N]
  const T* noise;       // replay normals, particle-major [N][pitch], or nullptr
  const double* rexp;   // replay unit exponentials [3][N], or nullptr
  const double* runif;  // replay uniforms of the discrete-time samplers [2N+1] (accept, flip, R gate)
  Control* ctl;
  unsigned long long* stats;  // [4]: #L, #F, #R, #cold of this attempt
  int64_t N;
  int64_t Npad;              // rows allocated: N rounded up to a multiple of 64
  int64_t first_pid;
  int D, pitch, CH, logG;
  int L;
  int iter;             // index of this attempt inside the current mjhmc_iterate call
  // FUSED kernels: n_fuse consecutive sampling iterations per particle in one launch, the state staying in
  // registers / LDS between them (iteration it uses RNG tick key.tick + it and stats[4 * it ...]).
  // xiter != nullptr: X after iteration it is also recorded at xiter + it * xiter_stride (ring slots) and
  // its dwelling times at dwell_ring + it * Npad; X_out is then not written (the last slot is the live state).
  int defer_r;          // != 0: R-movers keep their old momentum here (always 0 since round 5: their refresh is the jump-process kernel's)
  int n_fuse;
  int ab;               // host-side launch A/B flags (kAb*): always 0 in the shipped library, set by the test build's switches
  T* xiter;
  size_t xiter_stride;  // elements of T between consecutive ring slots
  int mode;             // kModeMJHMC / kModeControl / kModeCT
  T eps, chalf;         // epsilon and -epsilon/2 (hmc_state.py:88-91)
  T r_keep, r_mix;      // sqrt(1-beta), sqrt(beta) of HMCState.R (hmc_state.py:125-126)
  double p_r, p_flip;
  RngKey key;
};

template <typename T>
struct EvalArgs {
  const T* X;     // [n][pitch]
  T* G;           // [n][pitch] or nullptr
  T* E;           // [n] or nullptr
  T* EV;          // [n] or nullptr (kinetic energy of V when V != nullptr)
  const T* V;     // optional
  T* V_out;       // when non-null: V is GENERATED (tick-0 normals) and written here
  int64_t N;
  int64_t first_pid;
  int D, pitch, CH, logG;
  RngKey key;
};

// stand-alone leapfrog operator (see mjhmc_leap_kernel)
template <typename T>
struct LeapArgs {
  const T* X;      // [n][pitch]
  const T* V;
  T* X_out;        // [n][pitch]
  T* V_out;
  T* G;            // dE/dX at the end point, or nullptr
  T* EX;           // [n] or nullptr
  T* EV;
  int64_t N;
  int D, pitch, CH, logG;
  int L;
  T eps, chalf;
};

// Round 5: a MarkovJumpHMC iteration of a big batch with several particles per wavefront as TWO launches --
// mjhmc_traj_kernel integrates (the L proposal of every particle, and beside it the inverse-L proposal of the listed
// cold-cache particles), mjhmc_decide_kernel runs the jump process (see there).
template <typename T>
struct TrajArgs {
  const T* X_in;      // [Npad][pitch] pre-move state
  const T* V_in;
  T* X_out;           // the end point of L, written for every particle as if the move were taken
  T* V_out;
  T* EX_out;          // [Npad] its energies
  T* EV_out;
  T* Hwork;           // [Npad] H() of the inverse-L proposal F L F, written for the listed particles only
  const int* list;    // the particles with a cold cache
  const int* count;   // how many -- or NULL: n_listed then (a list carried over from the previous call, its length known to the host)
  const Control* ctl;
  int64_t N;
  int D, pitch, CH, logG;
  int L;
  int n_listed;
  int inv_blocks;     // leading workgroups of the grid that walk the list (the rest: one forward slot each)
  int rows;           // energies with a row form: a lane per particle (mjhmc_traj_rows_kernel); 0 = the group form (A/B)
  T eps, chalf;
};

template <typename T>
struct JumpDecideArgs {
  const T* X_in;      // pre-move state: what an F / R mover keeps
  const T* V_in;
  T* X_out;           // in: the L proposal; out: the successor
  T* V_out;
  const T* EX_in;
  const T* EV_in;
  T* EX_out;          // in: energies of the L proposal
  T* EV_out;
  const T* Hflf_in;   // H() of the cached inverse-L state, NaN = cold: then Hwork has it
  const T* Hwork;
  T* Hflf_out;
  double* dwell;
  double* dwell_ring;
  uint8_t* trans;
  int* next_list;     // the next iteration's cold caches: this iteration's F- and R-movers
  int* next_count;
  Control* ctl;
  unsigned long long* stats;   // [4]: #L, #F, #R (the cold tally is the list's length)
  int64_t N;
  int64_t first_pid;
  int D, pitch, CH, logG;
  int iter;
  T r_keep, r_mix;
  double p_r;
  RngKey key;
};

// ------------------------------------------------------------------------------------------
// building blocks
// ------------------------------------------------------------------------------------------

template <typename T, int E>
__device__ __forceinline__ void load_row(const T* row, const LaneMap& m, T (&r)[E]) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::n;
#pragma unroll
  for (int c = 0; c < E / VEC; ++c) {
    const int chunk = c * m.G + m.j;
    V v;
    if (chunk < m.CH) {
      v = *reinterpret_cast<const V*>(row + (size_t)chunk * VEC);
    } else {
      if constexpr (VEC == 2) v = V{0, 0};
      else v = V{0, 0, 0, 0};
    }
    if constexpr (VEC == 2) {
      r[c * 2] = v.x;
      r[c * 2 + 1] = v.y;
    } else {
      r[c * 4] = v.x;
      r[c * 4 + 1] = v.y;
      r[c * 4 + 2] = v.z;
      r[c * 4 + 3] = v.w;
    }
  }
}

template <typename T, int E>
__device__ __forceinline__ void store_row(T* row, const LaneMap& m, const T (&r)[E]) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::n;
#pragma unroll
  for (int c = 0; c < E / VEC; ++c) {
    const int chunk = c * m.G + m.j;
    if (chunk < m.CH) {
      V v;
      if constexpr (VEC == 2) v = V{r[c * 2], r[c * 2 + 1]};
      else v = V{r[c * 4], r[c * 4 + 1], r[c * 4 + 2], r[c * 4 + 3]};
      *reinterpret_cast<V*>(row + (size_t)chunk * VEC) = v;
    }
  }
}

// per-wave LDS stripe [C][64] of 16-byte chunks
template <typename T, int E>
__device__ __forceinline__ void stash_put(typename VecOf<T>::type (*st)[64], int lane, const T (&r)[E]) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::n;
#pragma unroll
  for (int c = 0; c < E / VEC; ++c) {
    if constexpr (VEC == 2) st[c][lane] = V{r[c * 2], r[c * 2 + 1]};
    else st[c][lane] = V{r[c * 4], r[c * 4 + 1], r[c * 4 + 2], r[c * 4 + 3]};
  }
}

template <typename T, int E>
__device__ __forceinline__ void stash_get(typename VecOf<T>::type (*st)[64], int lane, T (&r)[E]) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::n;
#pragma unroll
  for (int c = 0; c < E / VEC; ++c) {
    const V q = st[c][lane];
    if constexpr (VEC == 2) {
      r[c * 2] = q.x;
      r[c * 2 + 1] = q.y;
    } else {
      r[c * 4] = q.x;
      r[c * 4 + 1] = q.y;
      r[c * 4 + 2] = q.z;
      r[c * 4 + 3] = q.w;
    }
  }
}

// c * dE/dx_d: the half-kick product of hmc_state.py:88,91
template <class En, typename T, int E, class Ctx, class Lc>
__device__ __forceinline__ T kick_product(const En& en, T c, T xe, int e, int d, const Ctx& ctx, const Lc& lc) {
  if constexpr (En::kLinearIso) return xe * (c * en.inv_s2);
  else return c * en.template grad<E>(xe, e, d, ctx, lc);
}

// v += c * dE/dx_d as one fused multiply-add
template <class En>
struct HasScaledKick {
  template <class U>
  static auto test(int) -> decltype(U::kScaledKick, char());
  template <class U>
  static long test(...);
  static constexpr bool value = sizeof(test<En>(0)) == 1;
};

template <class En, typename T, int E, class Ctx, class Lc>
__device__ __forceinline__ T kick_fma(const En& en, T c, T xe, T ve, int e, int d, const Ctx& ctx, const Lc& lc) {
  if constexpr (En::kLinearIso) return __builtin_fma(xe, c * en.inv_s2, ve);
  else if constexpr (HasScaledKick<En>::value) return en.template kick<E>(c, xe, ve, e, d, ctx);  // force = xe * scale(ctx) for most elements
  else return __builtin_fma(c, en.template grad<E>(xe, e, d, ctx, lc), ve);
}

// M leapfrog steps, in place (hmc_state.py:86-100).
// EXACT = true (the replay kernels, i.e. the golden-vector path): the reference's literal operation order --
// half kicks NOT merged, every product rounded before its sum (the library is built with -ffp-contract=off);
// c*g of the closing kick is reused by the next opening kick (same product).  For forces that are exact
// products this reproduces NumPy's X and V bit for bit.
// EXACT = false (counter-RNG kernels): the same integrator with the two half kicks between consecutive steps
// merged into one (what the reference's compiled variant does, fast/hmc.py:6-98) and every update a single
// fused multiply-add: 2 + grad instead of 5 + grad vector instructions per element and step.  Differs from the
// literal order by rounding only (~1e-16 relative per step, inside the 1e-10 parity bar; the tests compare it
// with the oracle's literal order on the same Philox streams).
template <class En, typename T, int E, bool EXACT>
__device__ __forceinline__ void trajectory(const En& en, const typename En::template Local<E>& lc, const LaneMap& m,
                                           T (&x)[E], T (&v)[E], int L, T eps, T chalf) {
  if constexpr (EXACT) {
    T cg[E];
    {
      const auto ctx = en.prep(x, m);
#pragma unroll
      for (int e = 0; e < E; ++e) cg[e] = kick_product<En, T, E>(en, chalf, x[e], e, dim_of<T, E>(m, e), ctx, lc);
    }
    for (int s = 0; s < L; ++s) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        v[e] = v[e] + cg[e];
        x[e] = x[e] + eps * v[e];
      }
      const auto ctx = en.prep(x, m);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        cg[e] = kick_product<En, T, E>(en, chalf, x[e], e, dim_of<T, E>(m, e), ctx, lc);
        v[e] = v[e] + cg[e];
      }
    }
  } else {
    if (L <= 0) return;
    {
      const auto ctx = en.prep(x, m);
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = kick_fma<En, T, E>(en, chalf, x[e], v[e], e, dim_of<T, E>(m, e), ctx, lc);
    }
    const T cfull = chalf + chalf;
    for (int s = 0; s < L; ++s) {
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = __builtin_fma(eps, v[e], x[e]);
      const auto ctx = en.prep(x, m);
      const T c = (s == L - 1) ? chalf : cfull;  // the closing kick of the trajectory is a half kick
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = kick_fma<En, T, E>(en, c, x[e], v[e], e, dim_of<T, E>(m, e), ctx, lc);
    }
  }
}

template <typename T, int E>
__device__ __forceinline__ T kinetic(const T (&v)[E], const LaneMap& m) {
  T s = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) s = __builtin_fma(v[e], v[e], s);
  return group_sum(s, m.G) / T(2);  // hmc_state.py:50
}

// EV = kinetic(v) and EX = en.energy(x) of one state, the same bits as the two calls; energies that are a scaled group
// sum (kLaneEnergy) share one reduction ladder with the kinetic energy
template <class En, typename = void>
struct HasLaneEnergy {
  static constexpr bool value = false;
};
template <class En>
struct HasLaneEnergy<En, decltype((void)En::kLaneEnergy)> {
  static constexpr bool value = true;
};
template <class En, typename T, int E, class LC>
__device__ __forceinline__ void state_energies(const En& en, const LC& lc, const LaneMap& m, const T (&x)[E], const T (&v)[E],
                                               T& EX, T& EV) {
  if constexpr (HasLaneEnergy<En>::value) {
    T sv = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) sv = __builtin_fma(v[e], v[e], sv);
    const T sx = en.template energy_lane<E>(x, m, lc);
    T tv, tx;
    group_sum_pair(sv, sx, m.G, tv, tx);
    EV = tv / T(2);
    EX = en.energy_scale(tx);
  } else {
    EV = kinetic<T, E>(v, m);
    EX = en.energy(x, m, lc);
  }
}

// momentum-refresh noise for this lane's dims (hmc_state.py:125): replayed or counter RNG
template <typename T, int E>
__device__ __forceinline__ void refresh_noise(const T* noise_row, const RngKey& key, uint32_t pid, const LaneMap& m,
                                              T (&z)[E]) {
  constexpr int VEC = VecOf<T>::n;
  if (noise_row) {
    load_row<T, E>(noise_row, m, z);
    return;
  }
#pragma unroll
  for (int c = 0; c < E / VEC; ++c) {
    const int chunk = c * m.G + m.j;
#pragma unroll
    for (int h = 0; h < VEC / 2; ++h) {
      const int d = chunk * VEC + 2 * h;
      double z0 = 0.0, z1 = 0.0;
      if (chunk < m.CH && d < m.D) normal_pair(key, pid, (uint32_t)(d >> 1), z0, z1);
      z[c * VEC + 2 * h] = (T)z0;
      z[c * VEC + 2 * h + 1] = (d + 1 < m.D) ? (T)z1 : T(0);
    }
  }
}

// HMCState.R on the stashed momentum, chunk by chunk: v <- v*sqrt(1-beta) + n*sqrt(beta)
// (hmc_state.py:125-126).  The chunk loop is deliberately NOT unrolled: one Box-Muller pair's
// temporaries at a time keeps this rare branch from dictating the kernel's register budget.
template <typename T, int E, bool REPLAY>
__device__ __forceinline__ void refresh_stash(typename VecOf<T>::type (*st)[64], int lane, const T* noise_row,
                                              const RngKey& key, uint32_t pid, const LaneMap& m, T keep, T mix) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::n;
#pragma unroll 1
  for (int c = 0; c < E / VEC; ++c) {
    const int chunk = c * m.G + m.j;
    if (chunk >= m.CH) continue;  // stays zero
    const V v0 = st[c][lane];
    V z;
    if constexpr (REPLAY) {
      z = *reinterpret_cast<const V*>(noise_row + (size_t)chunk * VEC);
    } else {
      const int d = chunk * VEC;
      double z0, z1;
      normal_pair(key, pid, (uint32_t)(d >> 1), z0, z1);
      z.x = (T)z0;
      z.y = (d + 1 < m.D) ? (T)z1 : T(0);
      if constexpr (VEC == 4) {
        double z2 = 0.0, z3 = 0.0;
        if (d + 2 < m.D) normal_pair(key, pid, (uint32_t)((d + 2) >> 1), z2, z3);
        z.z = (T)z2;
        z.w = (d + 3 < m.D) ? (T)z3 : T(0);
      }
    }
    V r;
    r.x = v0.x * keep + z.x * mix;
    r.y = v0.y * keep + z.y * mix;
    if constexpr (VEC == 4) {
      r.z = v0.z * keep + z.z * mix;
      r.w = v0.w * keep + z.w * mix;
    }
    st[c][lane] = r;
  }
}

// exp(h) for |h| < 708 (normal result, no special cases): the library's range reduction and degree-11 polynomial
// without its overflow / underflow / NaN handling.  Within 1 ulp of libm (tools/check_device_math.hip).
// p = r * p + c with the constant c in a scalar register pair: one VOP3 instruction.  hipcc's own form of a Horner
// step is v_mov x2 (the 64-bit literal into the accumulator VGPR) + v_fmac: three vector instructions.
__device__ __forceinline__ double horner_sc(double r, double p, double c) {
  asm("v_fma_f64 %0, %1, %0, %2" : "+v"(p) : "v"(r), "s"(c));
  return p;
}

template <bool FENCED = true, bool SC = !FENCED>
__device__ __forceinline__ double exp_normal(double h) {
  if constexpr (FENCED) __builtin_amdgcn_sched_barrier(0);  // chain kept compact inside decide(): measured faster
  const double n = __builtin_rint(h * __longlong_as_double(0x3ff71547652b82feLL));        // h / ln 2
  double r = __builtin_fma(n, __longlong_as_double(0xbfe62e42fefa39efLL), h);             // - n ln2 (hi, lo)
  r = __builtin_fma(n, __longlong_as_double(0xbc7abc9e3b39803fLL), r);
  if constexpr (SC) {  // the funnel's force: this chain runs once per leapfrog step (decide()'s form below was
                            // measured faster there as it stands)
    double q = r * __longlong_as_double(0x3e5ade156a5dcb37LL) + __longlong_as_double(0x3e928af3fca7ab0cLL);
    q = horner_sc(r, q, __longlong_as_double(0x3ec71dee623fde64LL));
    q = horner_sc(r, q, __longlong_as_double(0x3efa01997c89e6b0LL));
    q = horner_sc(r, q, __longlong_as_double(0x3f2a01a014761f6eLL));
    q = horner_sc(r, q, __longlong_as_double(0x3f56c16c1852b7b0LL));
    q = horner_sc(r, q, __longlong_as_double(0x3f81111111122322LL));
    q = horner_sc(r, q, __longlong_as_double(0x3fa55555555502a1LL));
    q = horner_sc(r, q, __longlong_as_double(0x3fc5555555555511LL));
    q = horner_sc(r, q, __longlong_as_double(0x3fe000000000000bLL));
    q = __builtin_fma(r, q, 1.0);
    q = __builtin_fma(r, q, 1.0);
    const double yq = __builtin_amdgcn_ldexp(q, (int)n);
    if constexpr (FENCED) __builtin_amdgcn_sched_barrier(0);
    return yq;
  }
  double p = r * __longlong_as_double(0x3e5ade156a5dcb37LL) + __longlong_as_double(0x3e928af3fca7ab0cLL);
  p = __builtin_fma(r, p, __longlong_as_double(0x3ec71dee623fde64LL));
  p = __builtin_fma(r, p, __longlong_as_double(0x3efa01997c89e6b0LL));
  p = __builtin_fma(r, p, __longlong_as_double(0x3f2a01a014761f6eLL));
  p = __builtin_fma(r, p, __longlong_as_double(0x3f56c16c1852b7b0LL));
  p = __builtin_fma(r, p, __longlong_as_double(0x3f81111111122322LL));
  p = __builtin_fma(r, p, __longlong_as_double(0x3fa55555555502a1LL));
  p = __builtin_fma(r, p, __longlong_as_double(0x3fc5555555555511LL));
  p = __builtin_fma(r, p, __longlong_as_double(0x3fe000000000000bLL));
  p = __builtin_fma(r, p, 1.0);
  p = __builtin_fma(r, p, 1.0);
  const double y = __builtin_amdgcn_ldexp(p, (int)n);
  if constexpr (FENCED) __builtin_amdgcn_sched_barrier(0);
  return y;
}
// exp() of the funnel force, branch-free: v_ldexp_f64 rounds into the subnormal range and overflows to inf by itself
// (inf for very negative x_0 is what turns a runaway chain into the non-finite abort), so clamping the argument to
// +-1100 and restoring NaN is all the special handling there is.
__device__ __forceinline__ double exp_any(double h) {
  const double y = exp_normal<false>(__builtin_fmin(__builtin_fmax(h, -1100.0), 1100.0));
  return (h != h) ? h : y;
}
__device__ __forceinline__ float exp_any(float h) { return expf(h); }

// exp(-x) for the funnel force, once per leapfrog step: the argument is clamped to +-1100 (v_ldexp_f64 then rounds into
// the subnormal range / overflows to inf by itself; inf for very negative x_0 is what turns a runaway chain into the
// non-finite abort) and the sign rides on the operand modifiers.  A NaN x_0 needs no care here: it is already in every
// energy term that contains x_0, so the rates are NaN and the attempt aborts whatever this returns.
__device__ __forceinline__ double exp_neg(double x) {
  const double xc = __builtin_fmin(__builtin_fmax(x, -1100.0), 1100.0);
  const double n = __builtin_rint(xc * __longlong_as_double(0xbff71547652b82feLL));       // -x / ln 2
  double r = __builtin_fma(n, __longlong_as_double(0xbfe62e42fefa39efLL), -xc);           // -x - n ln2 (hi, lo)
  r = __builtin_fma(n, __longlong_as_double(0xbc7abc9e3b39803fLL), r);
  double q = r * __longlong_as_double(0x3e5ade156a5dcb37LL) + __longlong_as_double(0x3e928af3fca7ab0cLL);
  q = horner_sc(r, q, __longlong_as_double(0x3ec71dee623fde64LL));
  q = horner_sc(r, q, __longlong_as_double(0x3efa01997c89e6b0LL));
  q = horner_sc(r, q, __longlong_as_double(0x3f2a01a014761f6eLL));
  q = horner_sc(r, q, __longlong_as_double(0x3f56c16c1852b7b0LL));
  q = horner_sc(r, q, __longlong_as_double(0x3f81111111122322LL));
  q = horner_sc(r, q, __longlong_as_double(0x3fa55555555502a1LL));
  q = horner_sc(r, q, __longlong_as_double(0x3fc5555555555511LL));
  q = horner_sc(r, q, __longlong_as_double(0x3fe000000000000bLL));
  q = __builtin_fma(r, q, 1.0);
  q = __builtin_fma(r, q, 1.0);
  return __builtin_amdgcn_ldexp(q, (int)n);
}
__device__ __forceinline__ float exp_neg(float x) { return expf(-x); }
__device__ __forceinline__ ExpNegK exp_neg_pinned() {
  ExpNegK K;
  const unsigned long long bits[15] = {0xc091300000000000ULL /* -1100 */, 0x4091300000000000ULL /* 1100 */,
                                       0xbff71547652b82feULL, 0xbfe62e42fefa39efULL, 0xbc7abc9e3b39803fULL,
                                       0x3e5ade156a5dcb37ULL, 0x3e928af3fca7ab0cULL, 0x3ec71dee623fde64ULL,
                                       0x3efa01997c89e6b0ULL, 0x3f2a01a014761f6eULL, 0x3f56c16c1852b7b0ULL,
                                       0x3f81111111122322ULL, 0x3fa55555555502a1ULL, 0x3fc5555555555511ULL,
                                       0x3fe000000000000bULL};
#pragma unroll
  for (int i = 0; i < 15; ++i) {
    K.k[i] = __longlong_as_double((long long)bits[i]);
    asm volatile("" : "+v"(K.k[i]));   // opaque: a value in a vector register, not a literal to be re-created at its uses
  }
  return K;
}
__device__ __forceinline__ double exp_neg_k(double x, const ExpNegK& K) {   // exp_neg(), operation for operation
  const double xc = __builtin_fmin(__builtin_fmax(x, K.k[0]), K.k[1]);
  const double n = __builtin_rint(xc * K.k[2]);
  double r = __builtin_fma(n, K.k[3], -xc);
  r = __builtin_fma(n, K.k[4], r);
  double q = r * K.k[5] + K.k[6];
#pragma unroll
  for (int i = 7; i < 15; ++i) q = __builtin_fma(r, q, K.k[i]);
  q = __builtin_fma(r, q, 1.0);
  q = __builtin_fma(r, q, 1.0);
  return __builtin_amdgcn_ldexp(q, (int)n);
}

// Transition rate exp(dH) ** .5 (markov_jump_hmc.py:341-347).  Where exp(dH) is a normal number the rate is
// evaluated as exp(dH / 2) in one polynomial pass (within 1 ulp of the two-step value).  Where exp(dH)
// overflows (-> inf -> the non-finite abort), is subnormal (its square root then has the reference's coarse
// rounding) or is 0, and for NaN, the literal two-step form runs: thresholds and special values are exactly
// those of the reference expression.
// SC: the polynomial's constants in scalar registers (one particle per wave: the kernel is bound by its vector
// instruction count, and hipcc's own Horner step is three vector instructions)
template <bool SC = false>
__device__ __forceinline__ double jump_rate(double dH) {
  if (!(dH > -708.0 && dH < 709.0)) return sqrt(exp(dH));
  return exp_normal<true, SC>(0.5 * dH);
}

__device__ __forceinline__ double wait_time(double rate, double e, bool& bad) {
  // utils.py:37-48: rate == 0 -> inf; finite -> exponential(scale=1/rate) == (1/rate)*std_exp; else error
  if (rate == 0.0) return __builtin_huge_val();
  if (!isfinite(rate)) {
    bad = true;
    return __builtin_nan("");
  }
  return (1.0 / rate) * e;
}

// np.argmin over rows (first occurrence; a NaN wins as soon as it is met)
__device__ __forceinline__ int first_min3(double a, double b, double c) {
  int k = 0;
  double best = a;
  if (!(best != best)) {
    if (b < best || b != b) {
      k = 1;
      best = b;
    }
  }
  if (!(best != best)) {
    if (c < best || c != c) {
      k = 2;
      best = c;
    }
  }
  return k;
}

// ------------------------------------------------------------------------------------------
// the jump kernel (MJHMC mode)
// ------------------------------------------------------------------------------------------

// per-particle scalars travelling with a slot
template <typename T>
struct SlotScalars {
  T EX, EV, Hflf;  // Hflf is NaN <=> the inverse-L cache of this particle is cold
};

// pins the first use of a prefetched value to this program point (otherwise the scheduler may pull
// the use -- and with it a vmcnt wait that drains the whole prefetch -- up to the load)
__device__ __forceinline__ void use_here(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void use_here(float& v) { asm volatile("" : "+v"(v)); }

// Slot-relative row access.  A slot's 64/G particles are one contiguous span, so the address is
// (wave-uniform slot base) + (per-lane constant byte offset) + c * (uniform chunk stride): the
// per-slot address arithmetic is scalar, the vector part never changes.
template <typename T, int E, bool FULLROW>
__device__ __forceinline__ void slot_load(const char* base, uint32_t lane_off, uint32_t chunk_stride, const LaneMap& m,
                                          T (&r)[E]) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::n;
#pragma unroll
  for (int c = 0; c < E / VEC; ++c) {
    V v;
    if (FULLROW || c * m.G + m.j < m.CH) {
#if MJHMC_NT
      {
        using NV = T __attribute__((ext_vector_type(VecOf<T>::n)));
        const NV q = __builtin_nontemporal_load(reinterpret_cast<const NV*>(base + lane_off + c * chunk_stride));
        if constexpr (VEC == 2) v = V{q[0], q[1]};
        else v = V{q[0], q[1], q[2], q[3]};
      }
#else
      v = *reinterpret_cast<const V*>(base + lane_off + c * chunk_stride);
#endif
    } else {
      if constexpr (VEC == 2) v = V{0, 0};
      else v = V{0, 0, 0, 0};
    }
    if constexpr (VEC == 2) {
      r[c * 2] = v.x;
      r[c * 2 + 1] = v.y;
    } else {
      r[c * 4] = v.x;
      r[c * 4 + 1] = v.y;
      r[c * 4 + 2] = v.z;
      r[c * 4 + 3] = v.w;
    }
  }
}

template <typename T, int E, bool FULLROW>
__device__ __forceinline__ void slot_store(char* base, uint32_t lane_off, uint32_t chunk_stride, const LaneMap& m,
                                           const T (&r)[E]) {
  using V = typename VecOf<T>::type;
  constexpr int VEC = VecOf<T>::n;
#pragma unroll
  for (int c = 0; c < E / VEC; ++c) {
    if (FULLROW || c * m.G + m.j < m.CH) {
      V v;
      if constexpr (VEC == 2) v = V{r[c * 2], r[c * 2 + 1]};
      else v = V{r[c * 4], r[c * 4 + 1], r[c * 4 + 2], r[c * 4 + 3]};
#if MJHMC_NT
      {
        using NV = T __attribute__((ext_vector_type(VecOf<T>::n)));
        NV q;
        if constexpr (VEC == 2) q = NV{v.x, v.y};
        else q = NV{v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(q, reinterpret_cast<NV*>(base + lane_off + c * chunk_stride));
      }
#else
      *reinterpret_cast<V*>(base + lane_off + c * chunk_stride) = v;
#endif
    }
  }
}

// Rates, waiting-time draws and the first-minimum of one particle (markov_jump_hmc.py:366-396,
// utils.py:15-49).  All lanes of the particle's group enter with the same H values and leave with
// the same (k, dwell, bad).  When the group has >= 4 lanes the independent fp64 transcendental
// chains are evaluated ONCE, lane-parallel -- lane 0 works on the L clock, lane 1 on the R clock /
// the FLF rate, lane 2 on the F clock -- instead of three times in sequence in every lane
// (2 exp + 2 sqrt + 3 log + 3 div + 2 Philox  ->  1 of each), then exchanged inside the group.
// the unit exponential this lane contributes to decide() when the group has >= 4 lanes and the counter RNG is used
// (lane 0: L clock, lane 1: R clock, lanes 2..: F clock).  It depends on (particle id, tick) only, so a caller may
// draw it long before the energies exist (block-level decide: off the critical path between the two barriers).
__device__ __forceinline__ double decide_draw(const RngKey& key, const LaneMap& m, uint32_t pid) {
  const int role = m.j;
  const u32x4 w = philox4x32_10(pid, key.tick_lo, key.tick_hi, role == 1 ? kSlotExpR : kSlotExpLF, key.k0, key.k1);
  const double uA = u53(w.w0, w.w1);
  const double uF = group_lane(u53(w.w2, w.w3), m, 0);
  return neg_log_unit(role >= 2 ? uF : uA);
}

template <typename T, bool REPLAY, bool PREDRAWN = false>
__device__ __forceinline__ void decide(const JumpArgs<T>& a, const RngKey& key, const LaneMap& m, T H0, T HL, T Hflf,
                                       int64_t p, uint32_t pid, int& k, double& dwell, bool& bad, double e_pre = 0.0) {
  const double r_rate = a.p_r;
  double dL, dF, dR, l_rate, f_rate;
  if (m.G >= 4) {
    const int role = m.j;  // 0: L clock, 1: R clock (and the FLF rate), 2..: F clock
    const T dH = H0 - ((role == 1) ? Hflf : HL);
    // exp(H1 - H2) ** .5 (markov_jump_hmc.py:341-347)
    const double rate01 = (m.wpp == 1) ? jump_rate<true>((double)dH) : jump_rate<false>((double)dH);
    l_rate = group_lane(rate01, m, 0);
    const double flf_rate = group_lane(rate01, m, 1);
    const double mn = (flf_rate != flf_rate || l_rate != l_rate) ? __builtin_nan("") : fmin(flf_rate, l_rate);
    f_rate = flf_rate - mn;  // :368
    double e;
    if constexpr (PREDRAWN) {
      e = e_pre;
    } else if constexpr (REPLAY) {
      const int row = (role == 0) ? 0 : (role == 1 ? 2 : 1);
      e = a.rexp[(size_t)row * a.N + p];
    } else {
      if (m.wpp == 1) {
        // one particle per wave: the counter, key and slot are wave-uniform, so both Philox calls run
        // on the scalar unit (s_mul_i32 / s_mul_hi_u32) beside the vector work instead of in it
        const uint32_t spid = __builtin_amdgcn_readfirstlane(pid);
        const u32x4 w = philox4x32_10(spid, key.tick_lo, key.tick_hi, kSlotExpLF, key.k0, key.k1);
        const u32x4 q = philox4x32_10(spid, key.tick_lo, key.tick_hi, kSlotExpR, key.k0, key.k1);
        const double uL = u53(w.w0, w.w1), uF = u53(w.w2, w.w3), uR = u53(q.w0, q.w1);
        e = neg_log_unit(role == 0 ? uL : (role == 1 ? uR : uF));
      } else {
        e = decide_draw(key, m, pid);
      }
    }
    const double rate = (role == 0) ? l_rate : (role == 1 ? r_rate : f_rate);
    bool ignore = false;
    const double d = wait_time(rate, e, ignore);
    dL = group_lane(d, m, 0);
    dR = group_lane(d, m, 1);
    dF = group_lane(d, m, 2);
  } else {
    l_rate = jump_rate((double)(H0 - HL));
    __builtin_amdgcn_sched_barrier(0);  // keep the independent expansions from being interleaved:
    const double flf_rate = jump_rate((double)(H0 - Hflf));  // interleaved they cost ~90 extra VGPRs
    __builtin_amdgcn_sched_barrier(0);
    const double mn = (flf_rate != flf_rate || l_rate != l_rate) ? __builtin_nan("") : fmin(flf_rate, l_rate);
    f_rate = flf_rate - mn;
    double eL, eF, eR;
    if constexpr (REPLAY) {
      eL = a.rexp[p];
      eF = a.rexp[a.N + p];
      eR = a.rexp[2 * a.N + p];
    } else {
      const u32x4 w = philox4x32_10(pid, key.tick_lo, key.tick_hi, kSlotExpLF, key.k0, key.k1);
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 q = philox4x32_10(pid, key.tick_lo, key.tick_hi, kSlotExpR, key.k0, key.k1);
      __builtin_amdgcn_sched_barrier(0);
      eL = neg_log_unit(u53(w.w0, w.w1));
      __builtin_amdgcn_sched_barrier(0);
      eF = neg_log_unit(u53(w.w2, w.w3));
      __builtin_amdgcn_sched_barrier(0);
      eR = neg_log_unit(u53(q.w0, q.w1));
      __builtin_amdgcn_sched_barrier(0);
    }
    bool ignore = false;
    dL = wait_time(l_rate, eL, ignore);
    __builtin_amdgcn_sched_barrier(0);
    dF = wait_time(f_rate, eF, ignore);
    __builtin_amdgcn_sched_barrier(0);
    dR = wait_time(r_rate, eR, ignore);
    __builtin_amdgcn_sched_barrier(0);
  }
  // draw_from raises on a non-finite rate (utils.py:43-48); rate == 0 is fine (infinite wait)
  bad = !(isfinite(l_rate) && isfinite(f_rate) && isfinite(r_rate));
  k = first_min3(dL, dF, dR);
  dwell = (k == 0) ? dL : (k == 1 ? dF : dR);
}

// ContinuousTimeHMC (markov_jump_hmc.py:251-290): clocks FL (rate sqrt(exp(H0 - H_fl))), F (rate 1),
// R (rate p_r); min_idx is called with [f, fl, r], so ties go F, FL, R.  Returns k: 0 = FL, 1 = F, 2 = R.
template <typename T, bool REPLAY>
__device__ __forceinline__ void decide_ct(const JumpArgs<T>& a, const RngKey& key, const LaneMap& m, T H0, T HL,
                                          int64_t p, uint32_t pid, int& k, double& dwell, bool& bad) {
  const double fl_rate = jump_rate((double)(H0 - HL));
  const double r_rate = a.p_r;
  double dFL, dF, dR;
  if (m.G >= 4) {
    const int role = m.j;  // 0: FL clock, 1: R clock, 2..: F clock
    double e;
    if constexpr (REPLAY) {
      const int row = (role == 0) ? 0 : (role == 1 ? 2 : 1);
      e = a.rexp[(size_t)row * a.N + p];
    } else {
      const u32x4 w =
          philox4x32_10(pid, key.tick_lo, key.tick_hi, role == 1 ? kSlotExpR : kSlotExpLF, key.k0, key.k1);
      const double uA = u53(w.w0, w.w1);
      const double uF = group_lane(u53(w.w2, w.w3), m, 0);
      e = neg_log_unit(role >= 2 ? uF : uA);
    }
    const double rate = (role == 0) ? fl_rate : (role == 1 ? r_rate : 1.0);
    bool ignore = false;
    const double d = wait_time(rate, e, ignore);
    dFL = group_lane(d, m, 0);
    dR = group_lane(d, m, 1);
    dF = group_lane(d, m, 2);
  } else {
    double eFL, eF, eR;
    if constexpr (REPLAY) {
      eFL = a.rexp[p];
      eF = a.rexp[a.N + p];
      eR = a.rexp[2 * a.N + p];
    } else {
      const u32x4 w = philox4x32_10(pid, key.tick_lo, key.tick_hi, kSlotExpLF, key.k0, key.k1);
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 q = philox4x32_10(pid, key.tick_lo, key.tick_hi, kSlotExpR, key.k0, key.k1);
      __builtin_amdgcn_sched_barrier(0);
      eFL = neg_log_unit(u53(w.w0, w.w1));
      __builtin_amdgcn_sched_barrier(0);
      eF = neg_log_unit(u53(w.w2, w.w3));
      __builtin_amdgcn_sched_barrier(0);
      eR = neg_log_unit(u53(q.w0, q.w1));
      __builtin_amdgcn_sched_barrier(0);
    }
    bool ignore = false;
    dFL = wait_time(fl_rate, eFL, ignore);
    __builtin_amdgcn_sched_barrier(0);
    dF = wait_time(1.0, eF, ignore);
    __builtin_amdgcn_sched_barrier(0);
    dR = wait_time(r_rate, eR, ignore);
    __builtin_amdgcn_sched_barrier(0);
  }
  bad = !(isfinite(fl_rate) && isfinite(r_rate));
  const int kk = first_min3(dF, dFL, dR);  // rows f, fl, r (markov_jump_hmc.py:271)
  k = (kk == 0) ? 1 : (kk == 1 ? 0 : 2);
  dwell = (kk == 0) ? dF : (kk == 1 ? dFL : dR);
}

// REPLAY = true: random numbers come from host-supplied arrays (parity tests against recorded
// reference runs).  It is a compile-time switch because any global load consumed inside the slot
// body forces an in-order vmcnt wait that would also drain the prefetch of the next slot.
// FULLROW = true: every lane's chunks are inside the row (pitch == G * E), no per-chunk predicates.
// MODE: which sampler family's iteration this is (kModeMJHMC / kModeControl / kModeCT).
// WPP = 1: G == 64 (a whole wavefront per particle) is a compile-time fact: the reduction ladder, the group
// exchanges (v_readlane) and the Philox calls (scalar unit) specialise on it.
// WPP = 3: G == 4 (a quad per particle, e.g. ndims 17..32 in float64): two-level DPP reductions without the run-time
// ladder, group exchanges as quad_perm broadcasts instead of ds_bpermute (C4 -3 %, isotropic 32 x 10^6 fused -8 %).
// WPP = 5 (FUSED MJHMC launches, G == 64): as WPP = 1, with the unit exponentials of all the launch's waiting-time
// draws made UP FRONT.  They depend on (particle id, tick) only, and the Philox + log chain behind them is ~130 of
// decide()'s ~190 vector instructions, which a wave-per-particle kernel otherwise spends on three useful lanes.  At the
// top of a slot the wave fills its own stripe of an LDS table [iteration][clock], every quad of lanes working on a
// different iteration -- 16 draws per vector pass instead of one -- and the iterations then run decide() from the
// table (same functions on the same inputs: identical bits); no barrier is involved.
// WPP = 6: the same with half a wavefront per particle (G == 32, two particles per slot).
// FUSED = true: the launch runs a.n_fuse (<= kMaxFuse) consecutive sampling iterations per particle.  The
// chains are independent, so between iterations nothing has to leave the wave: X, V, EX, EV, H_flf stay in
// registers / the LDS stash, HBM sees one read and one write of the state per LAUNCH instead of per
// iteration (plus the ring snapshots when samples are recorded).  Iteration `it` uses RNG tick key.tick + it
// and tallies into stats[4 * it ..]; the first iteration that meets a non-finite rate is reported through
// Control::inv_iter and the host re-runs the launch up to that iteration (the input buffers are untouched).
// waves per SIMD the register allocator is asked to make room for.  Vector-pipe-bound energies whose 8-element
// float64 instance sits a register or two above an occupancy step ask for that step (funnel: 169 -> 168 VGPRs = three
// waves per SIMD instead of two, C4 -3 %); everything else takes the build-wide value.
template <class En, typename T, int E, typename = void>
struct JumpWaves {
  static constexpr int value = MJHMC_JUMP_WAVES;
};
template <class En, typename T, int E>
struct JumpWaves<En, T, E, decltype((void)En::kWavesRow64)> {
  static constexpr int value = (sizeof(T) * E == 64 && sizeof(T) == 8) ? En::kWavesRow64 : MJHMC_JUMP_WAVES;
};
template <class En, typename T, int E, int MODE, bool REPLAY, bool FULLROW, int WPP = 0, bool FUSED = false>
__global__ __launch_bounds__(256, (JumpWaves<En, T, E>::value)) void mjhmc_jump_kernel(const JumpArgs<T> a, const En en) {
  static_assert(!(FUSED && REPLAY), "recorded random numbers are replayed one iteration per launch");
  if (a.ctl->failed) {  // an earlier attempt of this batch was rolled back: do nothing
    // ... unless the failure belongs to THIS fused launch (another workgroup met it first): this workgroup must still run,
    // report an earlier first failure of its own particles if it has one, and flush its tallies of the good iterations
    // (single-iteration launches too: a workgroup that starts after another one of ITS launch has raised the flag -- two
    // processes sharing the GPU delay each other's workgroups by more than a short kernel runs -- must still tally its cold
    // caches: the failed attempt's evaluations count, Distribution.E_count)
    if (!FUSED) {
      if (__hip_atomic_load(&a.ctl->failed_iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.iter) return;
    } else {
      const int first = 0x7fffffff - __hip_atomic_load(&a.ctl->inv_iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (first < a.iter) return;
    }
  }
  // Persistent waves: wave w handles slots w, w + W, w + 2W, ... (a slot = the 64/G particles one
  // wavefront works on).  The NEXT slot's X, V and scalars are loaded into a second register set
  // before the current slot's trajectories start, so HBM latency hides behind the fp64 work.
  constexpr bool BD = (WPP == 5 || WPP == 6);  // the waiting-time draws of a slot's iterations are made up front (clock table in LDS)
  constexpr int kBdPpw = WPP == 6 ? 2 : 1;     // particles per wave of the block-decide forms
  static_assert(!BD || (FUSED && MODE == kModeMJHMC), "block-level decide exists for fused MJHMC launches");
  const int logG = WPP == 6 ? 5 : ((WPP == 1 || BD) ? 6 : (WPP == 3 ? 2 : a.logG));
  const int G = 1 << logG;
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave in block, scalar
  const int ppw = 64 >> logG;
  const int64_t nslots = a.Npad >> (6 - logG);  // rows are padded to a multiple of 64 particles
  const int64_t W = (int64_t)gridDim.x * 4;
  const int64_t wave = (int64_t)blockIdx.x * 4 + wib;
  const int gi = lane >> logG;
  LaneMap m;
  m.j = lane & (G - 1);
  m.G = G;
  m.D = a.D;
  m.CH = a.CH;
  m.lane0 = lane & ~(G - 1);
  m.wpp = WPP == 6 ? 6 : (BD ? 1 : WPP);
  constexpr int VEC = VecOf<T>::n;
  const uint32_t lane_off = ((uint32_t)gi * a.pitch + m.j * VEC) * (uint32_t)sizeof(T);
  const uint32_t chunk_stride = (uint32_t)(G * VEC * sizeof(T));
  const size_t slot_bytes = (size_t)ppw * a.pitch * sizeof(T);
  const auto lc = en.template local<E>(m);
  unsigned nL = 0, nF = 0, nR = 0, nCold = 0;  // per-lane tallies (group leaders only)
  bool any_bad = false;
  const int n_it = FUSED ? a.n_fuse : 1;
  int first_bad = 0x7fffffff;  // FUSED: first iteration of this lane's particles that met a non-finite rate
  __shared__ unsigned fused_tally[FUSED ? kMaxFuse : 1][4];
  if constexpr (FUSED) {
    fused_tally[threadIdx.x >> 2][threadIdx.x & 3] = 0;  // 256 threads == kMaxFuse * 4
    __syncthreads();
  }
  using Vec = typename VecOf<T>::type;
  constexpr int C = E / VEC;
  __shared__ Vec stash[4][2][C][64];
  Vec(*stash_x)[64] = stash[wib][0];
  Vec(*stash_v)[64] = stash[wib][1];
  __shared__ double bd_e[BD ? kMaxFuse : 1][4 * kBdPpw][3];  // unit exponentials [iteration][particle of the block][L, R, F clock]

  T nx[E], nv[E];
  SlotScalars<T> ns;
  auto fetch = [&](int64_t slot) {
    const int64_t p = slot * ppw + gi;
    ns.EX = a.EX_in[p];  // scalars first: they are the oldest loads, so waiting for them later
    ns.EV = a.EV_in[p];  // never waits for the rows behind them
    ns.Hflf = a.Hflf_in[p];
    slot_load<T, E, FULLROW>((const char*)a.X_in + slot * slot_bytes, lane_off, chunk_stride, m, nx);
    slot_load<T, E, FULLROW>((const char*)a.V_in + slot * slot_bytes, lane_off, chunk_stride, m, nv);
  };
  // FUSED launches spend tens of microseconds per slot: its loads are issued at the top of the slot instead of
  // one slot ahead, which frees the second register set (one more wave per SIMD)
  if (!FUSED && wave < nslots) fetch(wave);

  // (BD: every wave fills the clock table of its OWN slot -- a slot has n_it pairs, enough to keep all sixteen quads of
  // the wave busy -- so the waves of a workgroup need no barrier and walk their slots independently like everyone else)
  const int64_t slot_first = wave;
#pragma unroll 1
  for (int64_t slot0 = slot_first; slot0 < nslots; slot0 += W) {
    const bool have = true;
    const int64_t slot = slot0;
    const int64_t p = slot * ppw + gi;
    const bool alive = have && p < a.N;
    if constexpr (FUSED) fetch(slot);
    // The slot's pre-move state (x0, v0) is parked in this wave's private LDS stripe (lane-linear
    // 16-byte chunks: conflict-free ds_write_b128 / ds_read_b128, no barrier -- every lane reads
    // back only what it wrote).  Registers then hold just the working trajectory and the
    // prefetched next slot.
    T x[E], v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      x[e] = nx[e];
      v[e] = nv[e];
    }
    if constexpr (!FUSED) {
      stash_put<T, E>(stash_x, lane, x);
      stash_put<T, E>(stash_v, lane, v);
    }
    T EX0 = ns.EX, EV0 = ns.EV, Hcached = ns.Hflf;
    use_here(EX0);
    use_here(EV0);
    use_here(Hcached);
    if constexpr (!FUSED)
      fetch(slot0 + W < nslots ? slot0 + W : slot0);  // prefetch (unconditional, so waits stay countable): in flight during everything below
    const uint32_t pid = (uint32_t)(a.first_pid + p);
    RngKey key = a.key;
    int k;
    double dwell = 0.0;
    T EXn, EVn, Hc;
    bool tally_cold = false, r_applied = false;
    if constexpr (BD) {
      // this wave's stripe of the table: [iteration][wave * kBdPpw + particle of the slot][clock]; written and read by this
      // wave only (LDS operations of a wave execute in order)
      LaneMap mq = m;   // a quad of lanes per (particle, iteration) pair: lane 0 L clock, 1 R clock, 2 F clock
      mq.j = lane & 3;
      mq.G = 4;
      mq.lane0 = lane & ~3;
      mq.wpp = 3;
      for (int r = lane >> 2; r - (lane >> 2) < kBdPpw * n_it; r += 16) {  // r = kBdPpw * iteration + particle of the slot
        const int it = r / kBdPpw, q = r % kBdPpw;
        RngKey kt = a.key;
        const uint32_t lo = kt.tick_lo + (uint32_t)it;
        kt.tick_hi += lo < kt.tick_lo ? 1u : 0u;
        kt.tick_lo = lo;
        const double e = decide_draw(kt, mq, (uint32_t)(a.first_pid + slot * kBdPpw + q));
        if (it < n_it && (lane & 3) < 3) bd_e[it][kBdPpw * wib + q][lane & 3] = e;
      }
    }
#pragma unroll 1
    for (int it = 0; it < n_it; ++it) {
    if constexpr (FUSED) {
      stash_put<T, E>(stash_x, lane, x);
      stash_put<T, E>(stash_v, lane, v);
    }
    const bool warm = Hcached == Hcached;  // cache_active (hmc_state.py:43-44) is carried as "H_flf is not NaN"
    const T H0 = EX0 + EV0;  // HMCState.H (hmc_state.py:80-84)

    // inverse-L proposal F L F; only H() of it is ever read (markov_jump_hmc.py:360,367)
    T Hflf = Hcached;
    if (MODE == kModeMJHMC && !warm) {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = -v[e];
      trajectory<En, T, E, REPLAY>(en, lc, m, x, v, a.L, a.eps, a.chalf);
      T ex, ev;
      state_energies<En, T, E>(en, lc, m, x, v, ex, ev);
      Hflf = ex + ev;
      stash_get<T, E>(stash_x, lane, x);
      stash_get<T, E>(stash_v, lane, v);
    }

    // forward proposal L
    trajectory<En, T, E, REPLAY>(en, lc, m, x, v, a.L, a.eps, a.chalf);
    T EXL, EVL;
    state_energies<En, T, E>(en, lc, m, x, v, EXL, EVL);
    const T HL = EXL + EVL;

    bool bad = false;
    Hc = (T)__builtin_nan("");
    tally_cold = false;
    r_applied = false;
    if constexpr (MODE == kModeMJHMC) {
      if constexpr (BD) {
        // lane 0 of the particle's group: L clock, lane 1: R clock, others: F clock
        const double e_pre = bd_e[it][kBdPpw * wib + (kBdPpw == 2 ? (lane >> 5) : 0)][m.j < 2 ? m.j : 2];
        decide<T, false, true>(a, key, m, H0, HL, Hflf, alive ? p : 0, pid, k, dwell, bad, e_pre);
      } else {
        decide<T, REPLAY>(a, key, m, H0, HL, Hflf, alive ? p : 0, pid, k, dwell, bad);
      }
      tally_cold = !warm;
      // successor state (markov_jump_hmc.py:399-410)
      if (k == 0) {  // L: proposal accepted; the pre-move state becomes the cached inverse-L state
        EXn = EXL;
        EVn = EVL;
        Hc = H0;
      } else if (k == 1) {  // F: flip the momentum; clear_flf_cache (:409-410)
        stash_get<T, E>(stash_x, lane, x);
        stash_get<T, E>(stash_v, lane, v);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = -v[e];
        EXn = EX0;
        EVn = EV0;
      } else {  // R: refresh the momentum (hmc_state.py:121-129)
        stash_get<T, E>(stash_x, lane, x);
        if (!REPLAY && a.defer_r) {  // (no caller sets it any more: rounds 1-4 left the refresh to a compacted pass)
          stash_get<T, E>(stash_v, lane, v);
          EVn = EV0;
        } else {
          refresh_stash<T, E, REPLAY>(stash_v, lane, a.noise + (size_t)(alive ? p : 0) * a.pitch, key, pid, m,
                                      a.r_keep, a.r_mix);
          stash_get<T, E>(stash_v, lane, v);
          EVn = kinetic<T, E>(v, m);
        }
        EXn = EX0;
      }
    } else if constexpr (MODE == kModeCT) {
      decide_ct<T, REPLAY>(a, key, m, H0, HL, alive ? p : 0, pid, k, dwell, bad);
      if (k == 0) {  // FL: leap, then flip (markov_jump_hmc.py:258,278)
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = -v[e];
        EXn = EXL;
        EVn = EVL;
      } else if (k == 1) {  // F
        stash_get<T, E>(stash_x, lane, x);
        stash_get<T, E>(stash_v, lane, v);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = -v[e];
        EXn = EX0;
        EVn = EV0;
      } else {  // R (:285-286)
        stash_get<T, E>(stash_x, lane, x);
        refresh_stash<T, E, REPLAY>(stash_v, lane, a.noise + (size_t)(alive ? p : 0) * a.pitch, key, pid, m,
                                    a.r_keep, a.r_mix);
        stash_get<T, E>(stash_v, lane, v);
        EXn = EX0;
        EVn = kinetic<T, E>(v, m);
      }
    } else {
      // Discrete-time control samplers (markov_jump_hmc.py:116-148): propose L then F, accept with
      // min(1, exp(H0 - H1)) (a NaN difference accepts, as `Ediff < 0` is False, :112-113), flip with
      // probability p_flip, then refresh EVERY particle's momentum when the batch-wide gate fires.
      double uacc, uflip, ugate;
      if constexpr (REPLAY) {
        const int64_t pp = alive ? p : 0;
        uacc = a.runif[pp];
        uflip = a.runif[a.N + pp];
        ugate = a.runif[2 * a.N];
      } else {
        const u32x4 q = philox4x32_10(pid, key.tick_lo, key.tick_hi, kSlotExpR, key.k0, key.k1);
        const u32x4 f = philox4x32_10(pid, key.tick_lo, key.tick_hi, kSlotFlip, key.k0, key.k1);
        const u32x4 g = philox4x32_10(0xFFFFFFFFu, key.tick_lo, key.tick_hi, kSlotFlip, key.k0, key.k1);
        uacc = u53(q.w2, q.w3);
        uflip = u53(f.w0, f.w1);
        ugate = u53(g.w2, g.w3);
      }
      const double dH = (double)(H0 - HL);
      const bool accept = !(dH < 0.0) || (uacc < exp(dH));
      const bool flip = uflip < a.p_flip;
      const bool gate = ugate < a.p_r;
      if (accept) {
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = -v[e];
        EXn = EXL;
        EVn = EVL;
      } else {
        stash_get<T, E>(stash_x, lane, x);
        stash_get<T, E>(stash_v, lane, v);
        EXn = EX0;
        EVn = EV0;
      }
      if (flip) {
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = -v[e];
      }
      r_applied = gate;
      if (gate) {  // state.R() on the whole batch (:138-141)
        stash_put<T, E>(stash_v, lane, v);
        refresh_stash<T, E, REPLAY>(stash_v, lane, a.noise + (size_t)(alive ? p : 0) * a.pitch, key, pid, m,
                                    a.r_keep, a.r_mix);
        stash_get<T, E>(stash_v, lane, v);
        EVn = kinetic<T, E>(v, m);
      }
      k = (accept ? 1 : 0) | (flip ? 2 : 0);
      // tallies: slot 0 = accepted & flipped (l_count), 1 = flipped only (f_count), 2 = R applied,
      // 3 = accepted only (fl_count)   (markov_jump_hmc.py:143-148)
      tally_cold = (accept && !flip);
    }
    any_bad |= (bad && alive);
    if constexpr (FUSED) {
      // bookkeeping of iteration `it`; the successor state becomes the next iteration's pre-move state
      if (bad && alive) first_bad = min(first_bad, it);
      const bool lead = alive && m.j == 0;
      if constexpr (BD && kBdPpw == 1) {
        // a wavefront per particle: the move is wave-uniform, the tallies are two LDS atomics of lane 0 (the ballots of the
        // general form below cost ~25 vector instructions per iteration to say "one" -- 6 % of C2's iteration)
        if (lead) {
          atomicAdd(&fused_tally[it][k], 1u);
          if (tally_cold) atomicAdd(&fused_tally[it][3], 1u);
        }
      } else {
      unsigned long long b0, b1, b2;
      if constexpr (MODE == kModeControl) {
        b0 = __ballot(lead && k == 3);
        b1 = __ballot(lead && k == 2);
        b2 = __ballot(lead && r_applied);
      } else {
        b0 = __ballot(lead && k == 0);
        b1 = __ballot(lead && k == 1);
        b2 = __ballot(lead && k == 2);
      }
      const unsigned long long b3 = __ballot(lead && tally_cold);
      if (lane == 0) {
        if (b0) atomicAdd(&fused_tally[it][0], (unsigned)__popcll(b0));
        if (b1) atomicAdd(&fused_tally[it][1], (unsigned)__popcll(b1));
        if (b2) atomicAdd(&fused_tally[it][2], (unsigned)__popcll(b2));
        if (b3) atomicAdd(&fused_tally[it][3], (unsigned)__popcll(b3));
      }
      }
      if (a.xiter && have) {  // sample ring: X and the dwelling times after every iteration
        slot_store<T, E, FULLROW>((char*)(a.xiter + (size_t)it * a.xiter_stride) + slot * slot_bytes, lane_off,
                                  chunk_stride, m, x);
        a.dwell_ring[(size_t)it * a.Npad + p] = dwell;
      }
      EX0 = EXn;
      EV0 = EVn;
      Hcached = Hc;
      key.tick_hi += (key.tick_lo == 0xFFFFFFFFu) ? 1u : 0u;
      key.tick_lo += 1u;
    }
    }  // fused iterations

    // Stores are unconditional on purpose: rows beyond N are padding (allocated, never read back),
    // and every lane of a group writes the same scalar to the same address.  With no store behind
    // a branch the compiler can COUNT them, so the wait for the prefetched loads at the loop end is
    // vmcnt(#stores) instead of vmcnt(0) -- the wave never sits waiting for HBM write acks.
    if (have) {
      if (!FUSED || !a.xiter)
        slot_store<T, E, FULLROW>((char*)a.X_out + slot * slot_bytes, lane_off, chunk_stride, m, x);
      slot_store<T, E, FULLROW>((char*)a.V_out + slot * slot_bytes, lane_off, chunk_stride, m, v);
      a.EX_out[p] = EXn;
      a.EV_out[p] = EVn;
      a.Hflf_out[p] = Hc;
      a.dwell[p] = dwell;
      if constexpr (!FUSED) a.dwell_ring[p] = dwell;
      a.trans[p] = (uint8_t)k;
    }
    if (!FUSED && alive && m.j == 0) {
      if constexpr (MODE == kModeControl) {
        nL += (k == 3);
        nF += (k == 2);
        nR += r_applied ? 1u : 0u;
      } else {
        nL += (k == 0);
        nF += (k == 1);
        nR += (k == 2);
      }
      nCold += tally_cold ? 1u : 0u;
    }
  }
  if constexpr (FUSED) {
    for (int o = 32; o > 0; o >>= 1) first_bad = min(first_bad, __shfl_xor(first_bad, o));
    if (first_bad != 0x7fffffff && lane == 0) {
      a.ctl->failed = 1;
      atomicMax(&a.ctl->inv_iter, 0x7fffffff - (a.iter + first_bad));
    }
    __syncthreads();
    const int it = threadIdx.x >> 2, c = threadIdx.x & 3;
    if (it < n_it) {
      const unsigned t = fused_tally[it][c];
      if (t) atomicAdd(&a.stats[4 * it + c], (unsigned long long)t);
    }
    return;
  }
  if (any_bad) {  // draw_from's ValueError (utils.py:43-48): the host rolls this attempt back
    a.ctl->failed_iter = a.iter;   // (first: whoever sees the flag sees which launch raised it)
    __threadfence();
    a.ctl->failed = 1;
  }

  // integer bookkeeping: l/f/r counts and the number of cold inverse-L caches of this attempt
  // (markov_jump_hmc.py:413-415; Distribution.E_count / dEdX_count, distributions.py:62-75)
  __shared__ unsigned tally[4][4];
  for (int o = 32; o > 0; o >>= 1) {
    nL += __shfl_xor(nL, o);
    nF += __shfl_xor(nF, o);
    nR += __shfl_xor(nR, o);
    nCold += __shfl_xor(nCold, o);
  }
  if (lane == 0) {
    tally[wib][0] = nL;
    tally[wib][1] = nF;
    tally[wib][2] = nR;
    tally[wib][3] = nCold;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const unsigned long long t =
        (unsigned long long)tally[0][threadIdx.x] + tally[1][threadIdx.x] + tally[2][threadIdx.x] + tally[3][threadIdx.x];
    if (t) atomicAdd(&a.stats[threadIdx.x], t);
  }
}

// ------------------------------------------------------------------------------------------
// Round 5.  A MarkovJumpHMC iteration of a big batch with several particles per wavefront (C4: 16 particles per wave)
// used to be three launches on one stream -- inverse-L pass over the compacted cold caches (a trajectory's latency on a
// few waves: 32 us of C4's 269), jump kernel, list scan -- with the jump process inside the jump kernel: its ~350
// instructions per wave run on the lanes of the particles' groups between the trajectory and the stores of every slot.
// Now two launches of two KINDS:
//
//   mjhmc_traj_kernel    integrates and nothing else: memory-bound (C4: 1.02 GB in 0.20 ms = 5.1 TB/s).  Leading
//                        workgroups walk the list (the inverse-L proposals, beside the forward slots instead of in front of
//                        them); every other workgroup is one forward slot: load, trajectory, energies, store the end point
//                        AS IF the move were taken.  No LDS, no decision code, no second register set.
//   mjhmc_decide_kernel  the jump process with ONE LANE PER PARTICLE (64 live lanes of a wave instead of 16 groups'
//                        worth): rates, clocks, first minimum, dwelling time, counters (markov_jump_hmc.py:366-415):
//                        vector-pipe-bound.  The workgroup's movers (7 % of C4) are compacted in LDS and finished by lane
//                        groups: pre-move position put back, momentum flipped or redrawn (HMCState.R, with its kinetic
//                        energy); and they ARE the next iteration's list: one atomic per workgroup, no list scan.
//
// The same device functions as the jump kernel (trajectory, state_energies, decide, refresh_stash, kinetic) on the same
// inputs: the same bits (tests/test_gpu_fused.py compares the paths).  C4: 0.269 -> 0.262 ms per iteration, 125 000
// particles 0.057 -> 0.051.  (mjhmc_step_kernel can carry the trajectories of one part of a batch beside the jump process
// of another; api.hip does not use that: measured, no gain -- the trajectories keep the vector pipe half busy themselves.)
// ------------------------------------------------------------------------------------------
template <class En, typename T, int E>
__device__ __forceinline__ void traj_block(const TrajArgs<T>& a, const En& en, int vblock) {
  const int G = 1 << a.logG;
  const int lane = threadIdx.x & 63;
  LaneMap m;
  m.j = (int)(threadIdx.x & (G - 1));
  m.G = G;
  m.D = a.D;
  m.CH = a.CH;
  m.lane0 = lane & ~(G - 1);
  m.wpp = 0;
  const auto lc = en.template local<E>(m);
  const int ppb = 256 >> a.logG;  // particles per workgroup
  if (vblock < a.inv_blocks) {
    // inverse-L proposal F L F of the listed particles; only H() of it is ever read (markov_jump_hmc.py:360,367)
    const int n_cold = a.count ? *a.count : a.n_listed;
    for (int64_t first = (int64_t)vblock * ppb; first < n_cold; first += (int64_t)a.inv_blocks * ppb) {
      const int64_t idx = first + (threadIdx.x >> a.logG);
      const bool live = idx < n_cold;
      const int64_t p = a.list[live ? idx : 0];
      T x[E], v[E];
      load_row<T, E>(a.X_in + (size_t)p * a.pitch, m, x);
      load_row<T, E>(a.V_in + (size_t)p * a.pitch, m, v);
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = -v[e];
      trajectory<En, T, E, false>(en, lc, m, x, v, a.L, a.eps, a.chalf);
      T ex, ev;
      state_energies<En, T, E>(en, lc, m, x, v, ex, ev);
      if (live && m.j == 0) a.Hwork[p] = ex + ev;
    }
    return;
  }
  const int64_t p_raw = (int64_t)(vblock - a.inv_blocks) * ppb + (threadIdx.x >> a.logG);
  const bool alive = p_raw < a.N;
  const int64_t p = alive ? p_raw : a.N - 1;
  T x[E], v[E];
  load_row<T, E>(a.X_in + (size_t)p * a.pitch, m, x);
  load_row<T, E>(a.V_in + (size_t)p * a.pitch, m, v);
  trajectory<En, T, E, false>(en, lc, m, x, v, a.L, a.eps, a.chalf);
  T EXL, EVL;
  state_energies<En, T, E>(en, lc, m, x, v, EXL, EVL);
  if (alive) {
    store_row<T, E>(a.X_out + (size_t)p * a.pitch, m, x);
    store_row<T, E>(a.V_out + (size_t)p * a.pitch, m, v);
    if (m.j == 0) {
      a.EX_out[p] = EXL;
      a.EV_out[p] = EVL;
    }
  }
}

// ------------------------------------------------------------------------------------------
// The trajectory launch in ROW FORM: a lane per particle (energies with kRowForm; float64 rows of up to 16 chunks held by
// groups of 2 or 4 lanes elsewhere -- ndims 9 ... 32, C4).
//
// A group of G lanes per particle repeats the per-PARTICLE work of a force G times: the funnel's exp(-x0) (~25 float64
// instructions), the group reduction and broadcasts, the index and loop arithmetic.  With the whole row in one lane they
// are paid once per particle and the element updates stay what they were: C4's leapfrog step goes from ~110 vector
// instructions per 16 particles to ~125 per 64, and the launch becomes what its bytes say it is -- HBM-bound.
//
// HBM keeps the particle-major rows every other kernel reads (the jump-process launch, uploads, downloads): a wavefront
// (= a workgroup) moves its 64 rows between HBM and registers through a 16 KB LDS tile.  HBM side: chunk q of the tile's
// 64 * CH 16-byte chunks belongs to lane q % 64 -- consecutive lanes, consecutive addresses.  LDS side: row r, chunk c
// sits at slot r * 16 + (c ^ (r & 15)), so that both the row-order side (16 lanes = one row) and the lane-owns-a-row side
// (16 lanes = 16 rows, the same chunk) spread over all banks.  The listed cold caches' rows are gathered the same way,
// 16 lanes per row.
//
// The same arithmetic as the group form, operation for operation: lane j's partial sums in its element order, the partials
// paired as group_sum pairs them (rows_sum), exp_neg on the same argument, the same multiply-adds on the elements -- the
// same bits (tests/test_gpu_fused.py compares the launch strategies).
// ------------------------------------------------------------------------------------------
template <class En, typename = void>
struct HasRowForm {
  static constexpr bool value = false;
};
template <class En>
struct HasRowForm<En, decltype((void)En::kRowForm)> {
  static constexpr bool value = true;
};

template <int G>
__device__ __forceinline__ constexpr int row_dim(int j, int e) {   // dim_of for float64 (two elements per chunk)
  return ((e / 2) * G + j) * 2 + (e % 2);
}

// leapfrog steps [s0, s1) of a trajectory of L steps, a row at a time (trajectory<.., EXACT = false>); the opening half kick
// belongs to the FIRST part.  Parts that cover [0, L) in order are the whole trajectory, operation for operation: the relay
// of the fused row kernel integrates a trajectory in four parts, on four waves.
template <class En, typename T, int E, int G, class XK = NoExpK>
__device__ __forceinline__ void trajectory_rows_part(const En& en, T (&x)[G][E], T (&v)[G][E], bool first, int s0, int s1, int L,
                                                     T eps, T chalf, const XK& xk = XK{}) {
  if (L <= 0) return;
  if (first) {
    const auto ctx = en.template prep_rows<E, G>(x, xk);
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
      for (int e = 0; e < E; ++e) v[j][e] = en.template kick<E>(chalf, x[j][e], v[j][e], e, row_dim<G>(j, e), ctx);
  }
  const T cfull = chalf + chalf;
  for (int s = s0; s < s1; ++s) {
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
      for (int e = 0; e < E; ++e) x[j][e] = __builtin_fma(eps, v[j][e], x[j][e]);
    const auto ctx = en.template prep_rows<E, G>(x, xk);
    const T c = (s == L - 1) ? chalf : cfull;
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
      for (int e = 0; e < E; ++e) v[j][e] = en.template kick<E>(c, x[j][e], v[j][e], e, row_dim<G>(j, e), ctx);
  }
}

template <class En, typename T, int E, int G>
__device__ __forceinline__ void trajectory_rows(const En& en, T (&x)[G][E], T (&v)[G][E], int L, T eps, T chalf) {
  trajectory_rows_part<En, T, E, G>(en, x, v, true, 0, L, L, eps, chalf);
}

// the same part of the same trajectory with the row on TWO lanes: lane h of a pair holds x[h G/2 + jj] as xh[jj]
// (prep_rows_pair: the pair's sums are the row's, bit for bit; everything else is per coordinate)
template <class En, typename T, int E, int G, class XK = NoExpK>
__device__ __forceinline__ void trajectory_pair_part(const En& en, T (&xh)[G / 2][E], T (&vh)[G / 2][E], int h, bool first, int s0,
                                                     int s1, int L, T eps, T chalf, const XK& xk = XK{}) {
  constexpr int GH = G / 2;
  if (L <= 0) return;
  if (first) {
    const auto ctx = en.template prep_rows_pair<E, G>(xh, h, xk);
#pragma unroll
    for (int jj = 0; jj < GH; ++jj)
#pragma unroll
      for (int e = 0; e < E; ++e) vh[jj][e] = en.template kick<E>(chalf, xh[jj][e], vh[jj][e], e, row_dim<G>(h * GH + jj, e), ctx);
  }
  const T cfull = chalf + chalf;
  for (int s = s0; s < s1; ++s) {
#pragma unroll
    for (int jj = 0; jj < GH; ++jj)
#pragma unroll
      for (int e = 0; e < E; ++e) xh[jj][e] = __builtin_fma(eps, vh[jj][e], xh[jj][e]);
    const auto ctx = en.template prep_rows_pair<E, G>(xh, h, xk);
    const T c = (s == L - 1) ? chalf : cfull;
#pragma unroll
    for (int jj = 0; jj < GH; ++jj)
#pragma unroll
      for (int e = 0; e < E; ++e) vh[jj][e] = en.template kick<E>(c, xh[jj][e], vh[jj][e], e, row_dim<G>(h * GH + jj, e), ctx);
  }
}

// kinetic_rows() of a row on two lanes (valid in both lanes of the pair)
template <typename T, int E, int G>
__device__ __forceinline__ T kinetic_pair(const T (&vh)[G / 2][E]) {
  constexpr int GH = G / 2;
  T part[GH];
#pragma unroll
  for (int jj = 0; jj < GH; ++jj) {
    T s = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) s = __builtin_fma(vh[jj][e], vh[jj][e], s);
    part[jj] = s;
  }
  const T mine = (GH == 1) ? part[0] : part[0] + part[GH - 1];
  return (mine + dpp_mov<0xB1>(mine)) / T(2);
}

template <typename T, int E, int G>
__device__ __forceinline__ T kinetic_rows(const T (&v)[G][E]) {
  T part[G];
#pragma unroll
  for (int j = 0; j < G; ++j) {
    T s = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) s = __builtin_fma(v[j][e], v[j][e], s);
    part[j] = s;
  }
  return rows_sum<T, G>(part) / T(2);
}

// order the wavefront's LDS traffic: the tile is written by one set of lanes and read by another.  LDS operations of a wave
// execute in order; what is needed is that the compiler keeps them in order and that the writes have landed -- NOT a
// fence: a release fence also waits for the wave's global stores (vmcnt(0): the X rows' way to HBM, 2-5 us, in front of
// the V rows' pass through the tile).
__device__ __forceinline__ void wave_lds_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// The tile traffic of the row-form kernels: 64 rows between HBM (row order: chunk q = i * 64 + lane of the tile's 64 * CH
// chunks, (row, col) = (q / CH, q % CH)) and the lanes that own them, through tile[64 * 16] (slot row * 16 + (col ^ (row & 15))).
// FULL: every row has all its chunks (ndims 31, 32 / 15, 16): no predicate anywhere, and i = P * m + k walks rows
// 16 m + RPI * k + lane / RC, so a row-order slot is one of P per-lane bases + m * 256.
template <typename T, int E, int LOGG, bool FULL>
struct RowTile {
  static_assert(sizeof(T) == 8, "float64 rows");
  static constexpr int G = 1 << LOGG, C = E / 2, RC = C * G;   // RC: chunks of a full row
  static_assert(RC == 8 || RC == 16, "rows of 8 or 16 chunks");
  static constexpr int RPI = 64 / RC, P = 16 / RPI;
  using V = double2;
  // the chunks on their way between HBM and the tile: ONE vector value, not an array (an array of 16 chunks indexed from
  // unrolled loops was left in scratch memory -- 512 bytes written and read back per wave and matrix)
  using Stage = double __attribute__((ext_vector_type(2 * RC)));
  V* tile;
  int CH, pitch;
  int64_t N;

  // Every phase forms its own addresses from an opaque copy of the lane index: formed once, the compiler carries 16 + 16 LDS
  // and 2 x 16 global addresses across the trajectory loop in scratch memory -- and each reload in the store phase is a
  // vmcnt(0) wait, i.e. it waits for the previous STORE to reach memory (16 round trips per matrix).
  static __device__ __forceinline__ int fresh_lane() {
    int l = threadIdx.x & 63;   // (the relay kernel's workgroups have four waves)
    asm volatile("" : "+v"(l));
    return l;
  }
  struct Walk {   // short rows: (row, col) stepped, not divided
    int row, col, drow, dcol, CH;
    __device__ __forceinline__ void step() {
      col += dcol;
      row += drow;
      if (col >= CH) {
        col -= CH;
        row += 1;
      }
    }
  };
  __device__ __forceinline__ Walk walk0(int l) const {
    Walk w;
    w.CH = CH;
    w.row = l / CH;
    w.col = l - w.row * CH;
    w.drow = 64 / CH;
    w.dcol = 64 - w.drow * CH;
    return w;
  }
  // HBM rows -> registers, still in row order.  All loads are issued before anything waits; the passes a short row does
  // not need re-read the tile's last chunk and are dropped.
  __device__ __forceinline__ void fetch(const T* M, int64_t base, Stage& t) const {
    const int l = fresh_lane();
    const char* tb = reinterpret_cast<const char*>(M + (size_t)base * pitch);
    const uint32_t last_chunk = (uint32_t)(64 * CH - 1) * 16u;
#pragma unroll
    for (int i = 0; i < RC; ++i) {
      uint32_t off = (uint32_t)(i * 64 + l) * 16u;
      if (!FULL) off = min(off, last_chunk);
      const V q = *reinterpret_cast<const V*>(tb + off);
      t[2 * i] = q.x;
      t[2 * i + 1] = q.y;
    }
  }
  // the listed particles' rows (p_lane: the particle whose row this lane will own), RC lanes per row
  __device__ __forceinline__ void fetch_listed(const T* M, int p_lane, Stage& t) const {
    const int l = fresh_lane();
    Walk w = walk0(l);
#pragma unroll
    for (int i = 0; i < RC; ++i) {
      const int pr = __shfl(p_lane, FULL ? w.row : min(w.row, 63));
      const V q = *reinterpret_cast<const V*>(M + (size_t)pr * pitch + w.col * 2);
      t[2 * i] = q.x;
      t[2 * i + 1] = q.y;
      w.step();
    }
  }
  // ... through the tile into the lane that owns the row: r[j] = what lane j of the particle's group holds elsewhere
  __device__ __forceinline__ void to_rows(const Stage& t, T (&r)[G][E]) const {
    const int l = fresh_lane();
    if constexpr (FULL) {
      const int h = l / RC, col = l % RC;
#pragma unroll
      for (int k = 0; k < P; ++k) {
        const int rk = RPI * k + h, slot = rk * 16 + (col ^ rk);
#pragma unroll
        for (int m = 0; m < RC / P; ++m) tile[slot + m * 256] = V{t[2 * (P * m + k)], t[2 * (P * m + k) + 1]};
      }
    } else {
      Walk w = walk0(l);
#pragma unroll
      for (int i = 0; i < RC; ++i) {
        if (i < CH) tile[w.row * 16 + (w.col ^ (w.row & 15))] = V{t[2 * i], t[2 * i + 1]};
        w.step();
      }
    }
    wave_lds_fence();
    read_own(r);
    wave_lds_fence();
  }
  // the lane's own row of the tile <-> registers
  __device__ __forceinline__ void read_own(T (&r)[G][E]) const {
    const int l = fresh_lane();
    const int mine = l * 16, key = l & 15;
#pragma unroll
    for (int c = 0; c < RC; ++c) {
      V q = tile[mine + (c ^ key)];
      if (!FULL && c >= CH) q = V{0, 0};
      r[c % G][(c / G) * 2] = q.x;
      r[c % G][(c / G) * 2 + 1] = q.y;
    }
  }
  __device__ __forceinline__ void write_own(const T (&r)[G][E]) const {
    const int l = fresh_lane();
    const int mine = l * 16, key = l & 15;
#pragma unroll
    for (int c = 0; c < RC; ++c) tile[mine + (c ^ key)] = V{r[c % G][(c / G) * 2], r[c % G][(c / G) * 2 + 1]};
  }
  // registers -> the tile -> registers in row order (from_rows), then HBM (store).  Both matrices of a state go through
  // the tile before the first store is issued: a wait for anything older than a store (a reloaded spill on some path of
  // the compiler's, say) is then a wait for that store's round trip to memory.
  __device__ __forceinline__ void from_rows(const T (&r)[G][E], Stage& t) const {
    write_own(r);
    wave_lds_fence();
    const int l = fresh_lane();
    if constexpr (FULL) {
      const int h = l / RC, col = l % RC;
#pragma unroll
      for (int k = 0; k < P; ++k) {
        const int rk = RPI * k + h, slot = rk * 16 + (col ^ rk);
#pragma unroll
        for (int m = 0; m < RC / P; ++m) {
          const V q = tile[slot + m * 256];
          t[2 * (P * m + k)] = q.x;
          t[2 * (P * m + k) + 1] = q.y;
        }
      }
    } else {
      Walk w = walk0(l);
#pragma unroll
      for (int i = 0; i < RC; ++i) {
        const V q = tile[min(w.row, 63) * 16 + (w.col ^ (w.row & 15))];
        t[2 * i] = q.x;
        t[2 * i + 1] = q.y;
        w.step();
      }
    }
    wave_lds_fence();
  }
  __device__ __forceinline__ void store(T* M, int64_t base, const Stage& t) const {
    const int l = fresh_lane();
    char* tb = reinterpret_cast<char*>(M + (size_t)base * pitch);
    const int live_rows = (int)min((int64_t)64, N - base);   // < 64 in the batch's last tile only
    if (FULL && live_rows == 64) {
#pragma unroll
      for (int i = 0; i < RC; ++i) *reinterpret_cast<V*>(tb + (uint32_t)(i * 64 + l) * 16u) = V{t[2 * i], t[2 * i + 1]};
    } else {
      Walk w = walk0(l);
#pragma unroll
      for (int i = 0; i < RC; ++i) {
        if ((FULL || i < CH) && w.row < live_rows) *reinterpret_cast<V*>(tb + (uint32_t)(i * 64 + l) * 16u) = V{t[2 * i], t[2 * i + 1]};
        w.step();
      }
    }
  }
};

template <class En, typename T, int E, int LOGG, bool FULL>
__global__ __launch_bounds__(64, 2) void mjhmc_traj_rows_kernel(const TrajArgs<T> a, const En en, int n_walk) {
  using RT = RowTile<T, E, LOGG, FULL>;
  using Stage = typename RT::Stage;
  constexpr int G = RT::G;
  __shared__ typename RT::V tile[64 * 16];
  if (a.ctl->failed) return;
  const RT rt{tile, FULL ? RT::RC : a.CH, a.pitch, a.N};
  const int lane = threadIdx.x;
  T x[G][E], v[G][E];
  if ((int)blockIdx.x < n_walk) {
    // inverse-L proposal F L F of the listed particles; only H() of it is ever read (markov_jump_hmc.py:360,367)
    const int n_cold = a.count ? *a.count : a.n_listed;
    for (int64_t first = (int64_t)blockIdx.x * 64; first < n_cold; first += (int64_t)n_walk * 64) {
      const int64_t idx = first + lane;
      const bool live = idx < n_cold;
      const int p = a.list[live ? idx : 0];
      {
        Stage tx, tv;
        rt.fetch_listed(a.X_in, p, tx);
        rt.fetch_listed(a.V_in, p, tv);
        rt.to_rows(tx, x);
        rt.to_rows(tv, v);
      }
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int e = 0; e < E; ++e) v[j][e] = -v[j][e];
      trajectory_rows<En, T, E, G>(en, x, v, a.L, a.eps, a.chalf);
      const T ev = kinetic_rows<T, E, G>(v);
      const T ex = en.template energy_rows<E, G>(x);
      if (live) a.Hwork[p] = ex + ev;
    }
    return;
  }
  const int64_t base = (int64_t)((int)blockIdx.x - n_walk) * 64;
  {
    Stage tx, tv;
    rt.fetch(a.X_in, base, tx);
    rt.fetch(a.V_in, base, tv);
    rt.to_rows(tx, x);
    rt.to_rows(tv, v);
  }
  trajectory_rows<En, T, E, G>(en, x, v, a.L, a.eps, a.chalf);
  const T EVL = kinetic_rows<T, E, G>(v);
  const T EXL = en.template energy_rows<E, G>(x);
  {
    Stage tx, tv;
    rt.from_rows(x, tx);
    rt.from_rows(v, tv);
    rt.store(a.X_out, base, tx);
    rt.store(a.V_out, base, tv);
  }
  if (base + lane < a.N) {
    a.EX_out[base + lane] = EXL;
    a.EV_out[base + lane] = EVL;
  }
}

// ------------------------------------------------------------------------------------------
// Fused MarkovJumpHMC iterations in row form: n_fuse sampling iterations of 64 particles per wavefront with the state in
// registers between them -- what mjhmc_jump_kernel<.., FUSED> does for a group of lanes per particle, with a lane per
// particle.  Per iteration a wave runs the inverse-L trajectory (whenever one of its 64 caches is cold: nearly always),
// the L trajectory, the jump process (decide() in its one-lane form, as the jump-process launch of the compacted path)
// and the successor selection; HBM sees the state once per launch (+ the ring snapshots when samples are recorded).
// With ~129 vector instructions per leapfrog step and 64 particles this is bound by the vector pipe at 0.7 of the time the
// two launches of the compacted path need for their bytes (C4: 0.19 against 0.27 ms per iteration).
// (Every wave pays the inverse-L trajectory for its few cold lanes.  Letting each lane run at its OWN iteration -- rounds of
// one trajectory: warm lanes take their L proposal and move on, cold lanes their inverse-L proposal -- removes that, was
// written twice, is bit-identical, and is slower: a round then also pays the jump process.  DESIGN.md section 8c.)
//
// The momentum refresh of the wave's R-movers (a few per iteration) is spread over all lanes: the movers' rows go into
// the tile, every lane redraws one CHUNK of one mover -- normal_pair(key, particle, chunk) and the two products and the
// sum of refresh_stash, the same bits -- and the movers read their rows back.
// One wavefront per SIMD (x, v and their pre-move copies: 390 registers, the copies in the accumulation half), persistent:
// a wave strides over the tiles and adds its tallies up in LDS.
// ------------------------------------------------------------------------------------------
template <class En, typename T, int E, int LOGG, bool FULL>
__global__ __launch_bounds__(64, 1) void mjhmc_fused_rows_kernel(const JumpArgs<T> a, const En en) {
  using RT = RowTile<T, E, LOGG, FULL>;
  using Stage = typename RT::Stage;
  using V = typename RT::V;
  constexpr int G = RT::G, RC = RT::RC;
  __shared__ V tile[64 * 16];
  __shared__ unsigned tally[kMaxFuse][4];
  __shared__ int rtab[64];
  if (a.ctl->failed) {   // an earlier launch of this call met a non-finite rate and was rolled back: do nothing ...
    // ... unless the failure belongs to THIS launch (another wave met it first): this wave must still run, report an
    // earlier first failure of its own particles if it has one, and flush its tallies (the failed attempt's evaluations count)
    const int first = 0x7fffffff - __hip_atomic_load(&a.ctl->inv_iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (first < a.iter) return;
  }
  const RT rt{tile, FULL ? RC : a.CH, a.pitch, a.N};
  const int lane = threadIdx.x;
  const int n_it = a.n_fuse;
#pragma unroll
  for (int i = 0; i < kMaxFuse * 4 / 64; ++i) (&tally[0][0])[i * 64 + lane] = 0;
  wave_lds_fence();
  int first_bad = 0x7fffffff;
  LaneMap m1;   // decide() with the three clocks in one lane
  m1.j = 0;
  m1.G = 1;
  m1.D = 1;
  m1.CH = 1;
  m1.lane0 = 0;
  m1.wpp = 0;
  const int64_t n_tiles = a.Npad >> 6;
#pragma unroll 1
  for (int64_t ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
    const int64_t base = ti * 64, p = base + lane;
    const bool alive = p < a.N;
    const int64_t pc = alive ? p : a.N - 1;   // padding rows: somebody's scalars, nothing of theirs is stored
    T x[G][E], v[G][E];
    {
      Stage tx, tv;
      rt.fetch(a.X_in, base, tx);
      rt.fetch(a.V_in, base, tv);
      rt.to_rows(tx, x);
      rt.to_rows(tv, v);
    }
    T EX0 = a.EX_in[pc], EV0 = a.EV_in[pc], Hcached = a.Hflf_in[pc];
    const uint32_t pid = (uint32_t)(a.first_pid + p);
    RngKey key = a.key;
    int k = 0;
    double dwell = 0.0;
    T EXn = EX0, EVn = EV0, Hc = Hcached;
#pragma unroll 1
    for (int it = 0; it < n_it; ++it) {
      T x0[G][E], v0[G][E];
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          x0[j][e] = x[j][e];
          v0[j][e] = v[j][e];
        }
      const bool warm = Hcached == Hcached;   // cache_active (hmc_state.py:43-44) is carried as "H_flf is not NaN"
      const T H0 = EX0 + EV0;                 // HMCState.H (hmc_state.py:80-84)
      // inverse-L proposal F L F of the cold caches; only H() of it is ever read (markov_jump_hmc.py:360,367).  (Stepping it
      // TOGETHER with the L proposal -- two independent dependency chains for one wave per SIMD -- needs more than the 256
      // registers vector instructions can address: 389 instead of 2 x 129 instructions per step, the rest register moves.)
      T Hflf = Hcached;
      if (__ballot(!warm) != 0ull) {
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
          for (int e = 0; e < E; ++e) v[j][e] = -v[j][e];
        trajectory_rows<En, T, E, G>(en, x, v, a.L, a.eps, a.chalf);
        const T ev = kinetic_rows<T, E, G>(v);
        const T ex = en.template energy_rows<E, G>(x);
        if (!warm) Hflf = ex + ev;
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
          for (int e = 0; e < E; ++e) {
            x[j][e] = x0[j][e];
            v[j][e] = v0[j][e];
          }
      }
      // forward proposal L
      trajectory_rows<En, T, E, G>(en, x, v, a.L, a.eps, a.chalf);
      const T EVL = kinetic_rows<T, E, G>(v);
      const T EXL = en.template energy_rows<E, G>(x);
      const T HL = EXL + EVL;
      bool bad = false;
      decide<T, false>(a, key, m1, H0, HL, Hflf, pc, pid, k, dwell, bad);
      // successor state (markov_jump_hmc.py:399-410)
      const bool isL = k == 0, isF = k == 1, isR = k == 2;
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          x[j][e] = isL ? x[j][e] : x0[j][e];
          v[j][e] = isL ? v[j][e] : (isF ? -v0[j][e] : v0[j][e]);
        }
      EXn = isL ? EXL : EX0;
      EVn = isL ? EVL : EV0;
      Hc = isL ? H0 : (T)__builtin_nan("");   // L: the pre-move state becomes the cached inverse-L state; F, R: clear_flf_cache
      const unsigned long long bR = __ballot(isR && alive);
      if (bR != 0ull) {   // HMCState.R (hmc_state.py:121-129) of the wave's R-movers, a chunk per lane
        // (round 6, as in the relay kernel: the lanes draw n sqrt(beta) into the tile, a mover reads its row back and forms
        // v sqrt(1 - beta) + n sqrt(beta) itself -- the old momenta never travel through LDS; the same bits)
        const int nR = (int)__popcll(bR);
        if (isR && alive) rtab[(int)__popcll(bR & ((1ull << lane) - 1ull))] = lane;
        wave_lds_fence();
        const int l = RT::fresh_lane();
        for (int w = l; w < nR * RC; w += 64) {
          const int q = w / RC, c = w % RC;
          if (FULL || c < rt.CH) {
            const int r = rtab[q];
            const int slot = r * 16 + (c ^ (r & 15));
            const int d = c * 2;
            double z0, z1;
            normal_pair(key, (uint32_t)(a.first_pid + base + r), (uint32_t)c, z0, z1);
            const T zy = (d + 1 < a.D) ? (T)z1 : T(0);
            tile[slot] = V{(T)z0 * a.r_mix, zy * a.r_mix};
          }
        }
        wave_lds_fence();
        if (isR && alive) {
          T zr[G][E];
          rt.read_own(zr);
#pragma unroll
          for (int j = 0; j < G; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) v[j][e] = v[j][e] * a.r_keep + zr[j][e];
          EVn = kinetic_rows<T, E, G>(v);
        }
        wave_lds_fence();
      }
      // bookkeeping of iteration `it`; the successor state becomes the next iteration's pre-move state
      if (bad && alive) first_bad = min(first_bad, it);
      const unsigned long long b0 = __ballot(alive && isL), b1 = __ballot(alive && isF), b3 = __ballot(alive && !warm);
      if (lane == 0) {
        tally[it][0] += (unsigned)__popcll(b0);
        tally[it][1] += (unsigned)__popcll(b1);
        tally[it][2] += (unsigned)__popcll(bR);
        tally[it][3] += (unsigned)__popcll(b3);
      }
      if (a.xiter) {  // sample ring: X and the dwelling times after every iteration
        Stage tx;
        rt.from_rows(x, tx);
        rt.store(a.xiter + (size_t)it * a.xiter_stride, base, tx);
        a.dwell_ring[(size_t)it * a.Npad + p] = dwell;
      }
      EX0 = EXn;
      EV0 = EVn;
      Hcached = Hc;
      key.tick_hi += (key.tick_lo == 0xFFFFFFFFu) ? 1u : 0u;
      key.tick_lo += 1u;
    }  // fused iterations
    {
      Stage tx, tv;
      if (!a.xiter) rt.from_rows(x, tx);
      rt.from_rows(v, tv);
      if (!a.xiter) rt.store(a.X_out, base, tx);
      rt.store(a.V_out, base, tv);
    }
    if (alive) {
      a.EX_out[p] = EXn;
      a.EV_out[p] = EVn;
      a.Hflf_out[p] = Hc;
      a.dwell[p] = dwell;
      a.trans[p] = (uint8_t)k;
    }
  }
  for (int o = 32; o > 0; o >>= 1) first_bad = min(first_bad, __shfl_xor(first_bad, o));
  if (first_bad != 0x7fffffff && lane == 0) {
    a.ctl->failed = 1;
    atomicMax(&a.ctl->inv_iter, 0x7fffffff - (a.iter + first_bad));
  }
  wave_lds_fence();
  for (int i = lane; i < n_it * 4; i += 64) {
    const unsigned t = (&tally[0][0])[i];
    if (t) atomicAdd(&a.stats[i], (unsigned long long)t);
  }
}

// ------------------------------------------------------------------------------------------
// The fused row kernel with the inverse-L trajectories of a WORKGROUP's cold caches pooled and relayed (round 6).
//
// mjhmc_fused_rows_kernel integrates the inverse-L proposal in all 64 lanes of a wave whenever one of its caches is cold --
// at C4's 7 % movers that is 99 % of the wave-iterations, 1.86 trajectories executed per trajectory the chain needs.  Here a
// workgroup is four such waves (one per SIMD, 256 particles).  Who is cold in iteration t + 1 is known when iteration t has
// decided (every move but L clears the cache, markov_jump_hmc.py:404-410): those lanes put their (x, v) rows into an LDS pool
// (a row per pool lane, lane-linear), ~18 of 256 at C4.  The pool's trajectories -- ONE wave's worth of work for the four
// waves -- are integrated in four parts, one part per wave: wave p runs leapfrog steps [p L / 4, (p + 1) L / 4) of the pool
// between parts p - 1 and p of its OWN L trajectory, takes the pool's state from LDS where wave p - 1 left it and leaves it
// there for wave p + 1; the last part ends with the energies, H(F L F z) per pool lane, which the cold lanes pick up after
// their own L trajectory, in front of decide().  Every wave then issues 1.25 trajectories per iteration and none waits:
//
//     wave 0:  [pool 0][own 0][own 1][own 2][own 3]        the pool's parts follow each other on four SIMDs while every
//     wave 1:  [own 0][pool 1][own 1][own 2][own 3]        wave's own trajectory goes on beside them; hand-overs are LDS
//     wave 2:  [own 0][own 1][pool 2][own 2][own 3]        flags a wave polls (no s_barrier: a barrier per part would make
//     wave 3:  [own 0][own 1][own 2][pool 3][own 3]        everybody wait for the wave that has the pool)
//
// The same device functions on the same inputs in the same order (trajectory_rows_part: parts that cover [0, L) ARE the
// trajectory), so the bits are those of mjhmc_fused_rows_kernel and of the single-iteration kernels (tests/test_gpu_fused.py,
// tools/fuzz_rows.py).  A pooled particle takes TWO lanes of the relaying wave (half a row each: trajectory_pair_part), which
// makes a part of the pool's trajectory shorter than the part of the wave's own it stands beside -- the next wave never
// waits for it -- and leaves the wave's own state where it is (64 registers of pool state instead of 128: a first form with
// a lane per pooled particle moved 256 registers in and out of the accumulation half per part and measured 0.231 ms against
// the one-wave kernel's 0.210; its four hand-overs, each ~2 500 cycles, stood one after the other).  More than 32 cold caches
// in a workgroup (a chain's first iteration, a high refresh rate): the lanes beyond the pool integrate in their own wave as before.
// Order of LDS traffic between waves: a wave's LDS operations execute in order; a writer waits for its data (lgkmcnt(0))
// before it raises the flag, a reader's data reads are issued after the flag's value has come back.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void lds_wait_ge(const int* flag, int target) {
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
// the relay's flag: parts of the pool's trajectory done so far (low word) and, riding along, the pool's row count (high
// word) -- the next wave gets both with the one LDS read it polls with
__device__ __forceinline__ int lds_wait_seg(const unsigned long long* flag, int target) {
  unsigned long long f;
  while ((int)(unsigned)(f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target)
    __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
  return (int)(f >> 32);
}
__device__ __forceinline__ void lds_seg_set(unsigned long long* flag, int done, int rows) {   // (one lane)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __hip_atomic_store(flag, (unsigned long long)(unsigned)done | ((unsigned long long)(unsigned)rows << 32), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_WORKGROUP);
}

constexpr int kRelayWaves = 4;
constexpr int kRelayMinL = 12;   // leapfrog steps from which the relay kernel is the faster fused row form (launch_fused_rows);
                                 // an energy may state its own (En::kRelayMinL)
template <class En, typename = void>
struct RelayMinL {
  static constexpr int value = kRelayMinL;
};
template <class En>
struct RelayMinL<En, decltype((void)En::kRelayMinL)> {
  static constexpr int value = En::kRelayMinL;
};
constexpr int kPoolRows = 32;   // pooled particles per workgroup and iteration: two lanes each

template <class En, typename T, int E, int LOGG, bool FULL>
__global__ __launch_bounds__(64 * kRelayWaves, 1) void mjhmc_fused_rows_relay_kernel(const JumpArgs<T> a, const En en) {
  using RT = RowTile<T, E, LOGG, FULL>;
  using Stage = typename RT::Stage;
  using V = typename RT::V;
  constexpr int G = RT::G, GH = G / 2, RC = RT::RC, W = kRelayWaves, PR = kPoolRows;
  __shared__ V tiles[W][64 * 16];
  __shared__ V pool[2 * RC * PR];      // chunk c of pool row r at [c * PR + r]: the x chunks, then the v chunks
  __shared__ T pool_H[PR];
  __shared__ unsigned long long relay_flag;   // lds_wait_seg / lds_seg_set
  __shared__ unsigned long long tally[kMaxFuse][2];   // per iteration (#L | #F << 32), (#R | #cold << 32)
  __shared__ int pool_n[2];            // rows asked for, by parity of the epoch
  __shared__ int arrived, go;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) {
    int g = 1;
    if (a.ctl->failed) {   // as mjhmc_fused_rows_kernel; decided once for the workgroup (its waves must agree)
      const int first = 0x7fffffff - __hip_atomic_load(&a.ctl->inv_iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (first < a.iter) g = 0;
    }
    go = g;
    arrived = 0;
    relay_flag = 0ull;
    pool_n[0] = 0;
    pool_n[1] = 0;
  }
  for (int i = threadIdx.x; i < kMaxFuse * 2; i += 64 * W) (&tally[0][0])[i] = 0ull;
  __syncthreads();
  if (!go) return;
  V* const tile = tiles[wave];
  const RT rt{tile, FULL ? RC : a.CH, a.pitch, a.N};
  const int n_it = a.n_fuse;
  int first_bad = 0x7fffffff;
  LaneMap m1;   // decide() with the three clocks in one lane
  m1.j = 0;
  m1.G = 1;
  m1.D = 1;
  m1.CH = 1;
  m1.lane0 = 0;
  m1.wpp = 0;
  const int L = a.L;
  // where a wave's own trajectory stops for its part of the pool's.  Part p of the pool's trajectory (L / 4 steps on two
  // lanes per particle + the hand-over through LDS: ~0.19 L + 1.7 own steps' worth of time, tools/rows_stamps.py) can start
  // when part p - 1 is done: wave p turns to it after own_stop(p) steps of its own, when that is about to be the case --
  // standing at L p / 4 it waited ~1 700 cycles per iteration at C4
  auto own_stop = [&](int p) { return p >= W ? L : min(L, (p * (19 * L + 170) + 50) / 100); };
  int epoch = 0;   // (tile, iteration) pairs this workgroup has started: the same sequence in its four waves
  [[maybe_unused]] bool stamp_on = false;   // (timing build only: ROWS_STAMP)
  [[maybe_unused]] int stamp_it = 0;
  // A wave's rows asked of the pool, in three steps that leave their LDS round trips behind other work: pool_alloc (the rows'
  // numbers: one returning atomic per wave, issued when decide() is through), pool_write (the rows, when the movers have
  // their new momenta; returns the lane's pool row -- >= PR: the pool is full, the lane integrates in its wave) and
  // pool_arrive (at the top of the next iteration, behind its register copies).
  unsigned long long dep_mask = 0ull;
  int dep_base = 0;
  auto pool_alloc = [&](bool cold, int par) {
    dep_mask = __ballot(cold);
    dep_base = 0;
    if (dep_mask != 0ull && lane == __ffsll((long long)dep_mask) - 1) dep_base = atomicAdd(&pool_n[par], (int)__popcll(dep_mask));
  };
  auto pool_write = [&](bool cold, const T (&x)[G][E], const T (&v)[G][E]) -> int {
    int sl = -1;
    if (dep_mask != 0ull) {
      const int b = __builtin_amdgcn_readlane(dep_base, __ffsll((long long)dep_mask) - 1);
      if (cold) {
        sl = b + (int)__popcll(dep_mask & ((1ull << lane) - 1ull));
        if (sl < PR) {
#pragma unroll
          for (int c = 0; c < RC; ++c) {
            // (the chunks of a row's second half stand 8 rows further round: the two lanes of a pair then read 128 bytes
            // apart in a bank row instead of on the same banks -- SQ_LDS_BANK_CONFLICT was 1.9 % of the kernel's cycles)
            const int at = c * PR + ((sl + 8 * ((c % G) / GH)) & (PR - 1));
            pool[at] = V{x[c % G][(c / G) * 2], x[c % G][(c / G) * 2 + 1]};
            pool[RC * PR + at] = V{v[c % G][(c / G) * 2], v[c % G][(c / G) * 2 + 1]};
          }
        }
      }
    }
    return sl;
  };
  auto pool_arrive = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) atomicAdd(&arrived, 1);
  };
  // part `ph` of the pool's inverse-L trajectories F L F (only H() of them is ever read: markov_jump_hmc.py:360,367), a
  // pooled particle on the lanes 2 r, 2 r + 1: ~115 vector instructions per leapfrog step instead of the ~155 of a whole row
  // per lane, and 64 registers of state beside the wave's own -- nothing of the wave's has to be moved out of the way
  auto relay = [&](int ph, const ExpNegK& xk) {
    ROWS_STAMP(2);
    const int par = epoch & 1;
    int n;
    if (ph == 0) {
      lds_wait_ge(&arrived, W * (epoch + 1));
      n = min(__hip_atomic_load(&pool_n[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), PR);
      if (lane == 0) pool_n[par ^ 1] = 0;   // the next epoch's requests start when this epoch's energies are out
    } else {
      n = lds_wait_seg(&relay_flag, W * epoch + ph);
    }
    n = __builtin_amdgcn_readfirstlane(n);
    ROWS_STAMP(3);
    if (n > 0) {
      T px[GH][E], pv[GH][E];
      const int l = RT::fresh_lane();
      const int h = l & 1, r = min(l >> 1, n - 1);   // (the lanes beyond the pool repeat its last row; they store nothing)
      const bool mine = (l >> 1) < n;
      const V* src = pool + h * GH * PR + ((r + 8 * h) & (PR - 1));
#pragma unroll
      for (int k = 0; k < E / 2; ++k)
#pragma unroll
        for (int jj = 0; jj < GH; ++jj) {
          const V qx = src[(k * G + jj) * PR], qv = src[(RC + k * G + jj) * PR];
          px[jj][2 * k] = qx.x;
          px[jj][2 * k + 1] = qx.y;
          pv[jj][2 * k] = (ph == 0) ? -qv.x : qv.x;
          pv[jj][2 * k + 1] = (ph == 0) ? -qv.y : qv.y;
        }
      trajectory_pair_part<En, T, E, G>(en, px, pv, h, ph == 0, ph * L / W, (ph + 1) * L / W, L, a.eps, a.chalf, xk);
      if (ph == W - 1) {
        const T ev = kinetic_pair<T, E, G>(pv);
        const T ex = en.template energy_pair<E, G>(px, h, xk);
        if (mine && h == 0) pool_H[r] = ex + ev;
      } else if (mine) {
        V* dst = pool + h * GH * PR + ((r + 8 * h) & (PR - 1));
#pragma unroll
        for (int k = 0; k < E / 2; ++k)
#pragma unroll
          for (int jj = 0; jj < GH; ++jj) {
            dst[(k * G + jj) * PR] = V{px[jj][2 * k], px[jj][2 * k + 1]};
            dst[(RC + k * G + jj) * PR] = V{pv[jj][2 * k], pv[jj][2 * k + 1]};
          }
      }
    }
    if (lane == 0) lds_seg_set(&relay_flag, W * epoch + ph + 1, n);
    ROWS_STAMP(4);
  };
  const int64_t n_wtiles = a.Npad >> 6, n_gtiles = (n_wtiles + W - 1) / W;
#pragma unroll 1
  for (int64_t gt = blockIdx.x; gt < n_gtiles; gt += gridDim.x) {
    const int64_t wt = gt * W + wave;
    const bool wave_live = wt < n_wtiles;   // (the batch's last workgroup tile may hold fewer than four wave tiles: the
                                            // others walk through the protocol with nothing alive)
    const int64_t base = wt * 64, p = base + lane;
    const bool alive = wave_live && p < a.N;
    const int64_t pc = alive ? p : a.N - 1;   // padding rows: somebody's scalars, nothing of theirs is stored
    T x[G][E], v[G][E];
    // the particle's scalars are asked for FIRST: behind the rows' way through the tile (whose fences the compiler does not
    // move loads across) their round trip to memory stood alone at the head of every tile -- a lone wave hides nothing
    stamp_on = gt == (int64_t)blockIdx.x;
    stamp_it = 0;
    ROWS_STAMP(13);
    T EX0 = a.EX_in[pc], EV0 = a.EV_in[pc], Hcached = a.Hflf_in[pc];
    if (wave_live) {
      Stage tx, tv;
      rt.fetch(a.X_in, base, tx);
      rt.fetch(a.V_in, base, tv);
      rt.to_rows(tx, x);
      rt.to_rows(tv, v);
    } else {
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          x[j][e] = T(0);
          v[j][e] = T(0);
        }
    }
    const uint32_t pid = (uint32_t)(a.first_pid + p);
    RngKey key = a.key;
    int k = 0;
    double dwell = 0.0;
    T EXn = EX0, EVn = EV0, Hc = Hcached;
    ROWS_STAMP(14);
    pool_alloc(alive && !(Hcached == Hcached), epoch & 1);
    int slot = pool_write(alive && !(Hcached == Hcached), x, v);
#pragma unroll 1
    for (int it = 0; it < n_it; ++it) {
      stamp_on = gt == (int64_t)blockIdx.x;
      stamp_it = it;
      ROWS_STAMP(0);
      T x0[G][E], v0[G][E];
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          x0[j][e] = x[j][e];
          v0[j][e] = v[j][e];
        }
      pool_arrive();   // (this iteration's cold rows were written before the copies above)
      const bool warm = Hcached == Hcached;   // cache_active (hmc_state.py:43-44) is carried as "H_flf is not NaN"
      const T H0 = EX0 + EV0;                 // HMCState.H (hmc_state.py:80-84)
      const bool pooled = !warm && slot >= 0 && slot < PR;
      const bool in_wave = !warm && alive && !pooled;
      T Hflf = Hcached;
      if (__ballot(in_wave) != 0ull) {   // the pool was full: as mjhmc_fused_rows_kernel
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
          for (int e = 0; e < E; ++e) v[j][e] = -v[j][e];
        trajectory_rows<En, T, E, G>(en, x, v, L, a.eps, a.chalf);
        const T ev = kinetic_rows<T, E, G>(v);
        const T ex = en.template energy_rows<E, G>(x);
        if (in_wave) Hflf = ex + ev;
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
          for (int e = 0; e < E; ++e) {
            x[j][e] = x0[j][e];
            v[j][e] = v0[j][e];
          }
      }
      ROWS_STAMP(1);
      // forward proposal L in four parts, this wave's part of the pool's trajectories between two of them
      const ExpNegK xk = exp_neg_pinned();   // (the force's exp(-x0) constants: vector registers for the length of the trajectories)
#pragma unroll 1
      for (int ph = 0; ph < W; ++ph) {
        if (wave == ph) relay(ph, xk);
        trajectory_rows_part<En, T, E, G>(en, x, v, ph == 0, own_stop(ph), own_stop(ph + 1), L, a.eps, a.chalf, xk);
      }
      ROWS_STAMP(5);
      const T EVL = kinetic_rows<T, E, G>(v);
      const T EXL = en.template energy_rows<E, G>(x);
      const T HL = EXL + EVL;
      ROWS_STAMP(6);
      (void)lds_wait_seg(&relay_flag, W * (epoch + 1));
      if (pooled) Hflf = pool_H[slot];
      ROWS_STAMP(7);
      bool bad = false;
      decide<T, false>(a, key, m1, H0, HL, Hflf, pc, pid, k, dwell, bad);
      ROWS_STAMP(8);
      // successor state (markov_jump_hmc.py:399-410)
      const bool isL = k == 0, isF = k == 1, isR = k == 2;
      const bool more = it + 1 < n_it;
      if (more) pool_alloc(alive && !isL, (epoch + 1) & 1);   // the next iteration's cold caches: every move but L
      if (!isL) {   // (the movers' lanes only: register moves under their execution mask instead of two selects per value in all lanes)
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
          for (int e = 0; e < E; ++e) {
            x[j][e] = x0[j][e];
            v[j][e] = isF ? -v0[j][e] : v0[j][e];
          }
      }
      EXn = isL ? EXL : EX0;
      EVn = isL ? EVL : EV0;
      Hc = isL ? H0 : (T)__builtin_nan("");   // L: the pre-move state becomes the cached inverse-L state; F, R: clear_flf_cache
      const unsigned long long bR = __ballot(isR && alive);
      ROWS_STAMP(9);
      if (bR != 0ull) {   // HMCState.R (hmc_state.py:121-129) of the wave's R-movers, a chunk per lane
        // The lanes draw n sqrt(beta) chunk by chunk into the tile; a mover reads its row of them back and forms
        // v sqrt(1 - beta) + n sqrt(beta) itself -- the two products and the sum of refresh_stash, the same bits --, so the
        // old momenta never travel: two LDS round trips in a lone wave's stream instead of four.
        const int nR = (int)__popcll(bR);
        const int l = RT::fresh_lane();
        constexpr int RPI = 64 / RC;       // movers per pass: lanes [g RC, (g + 1) RC) redraw the pass's g-th mover
        const int c = l % RC, g_l = l / RC;
        unsigned long long left = bR;      // (wave-uniform: the movers' lanes are picked off the ballot on the scalar unit)
        for (int q0 = 0; q0 < nR; q0 += RPI) {
          int r = 0;
#pragma unroll
          for (int g = 0; g < RPI; ++g) {
            const int rg = __ffsll((long long)left) - 1;
            left &= left - 1ull;
            r = (g_l == g) ? rg : r;
          }
          if (q0 + g_l < nR && (FULL || c < rt.CH)) {
            const int sl = r * 16 + (c ^ (r & 15));
            const int d = c * 2;
            double z0, z1;
            normal_pair(key, (uint32_t)(a.first_pid + base + r), (uint32_t)c, z0, z1);
            const T zy = (d + 1 < a.D) ? (T)z1 : T(0);
            tile[sl] = V{(T)z0 * a.r_mix, zy * a.r_mix};
          }
        }
        wave_lds_fence();
        if (isR && alive) {
          T zr[G][E];
          rt.read_own(zr);
#pragma unroll
          for (int j = 0; j < G; ++j)
#pragma unroll
            for (int e = 0; e < E; ++e) v[j][e] = v[j][e] * a.r_keep + zr[j][e];
          EVn = kinetic_rows<T, E, G>(v);
        }
        wave_lds_fence();
      }
      ROWS_STAMP(10);
      if (more) slot = pool_write(alive && !isL, x, v);   // (in front of the bookkeeping: its LDS writes land behind it)
      // bookkeeping of iteration `it`; the successor state becomes the next iteration's pre-move state
      if (bad && alive) first_bad = min(first_bad, it);
      const unsigned long long b0 = __ballot(alive && isL), b1 = __ballot(alive && isF), b3 = __ballot(alive && !warm);
      if (lane == 0) {
        atomicAdd(&tally[it][0], (unsigned long long)__popcll(b0) | ((unsigned long long)__popcll(b1) << 32));
        atomicAdd(&tally[it][1], (unsigned long long)__popcll(bR) | ((unsigned long long)__popcll(b3) << 32));
      }
      if (a.xiter && wave_live) {  // sample ring: X and the dwelling times after every iteration
        Stage tx;
        rt.from_rows(x, tx);
        rt.store(a.xiter + (size_t)it * a.xiter_stride, base, tx);
        a.dwell_ring[(size_t)it * a.Npad + p] = dwell;
      }
      EX0 = EXn;
      EV0 = EVn;
      Hcached = Hc;
      key.tick_hi += (key.tick_lo == 0xFFFFFFFFu) ? 1u : 0u;
      key.tick_lo += 1u;
      ++epoch;
      ROWS_STAMP(11);
      ROWS_STAMP(12);
    }  // fused iterations
    if (wave_live) {
      Stage tx, tv;
      if (!a.xiter) rt.from_rows(x, tx);
      rt.from_rows(v, tv);
      if (!a.xiter) rt.store(a.X_out, base, tx);
      rt.store(a.V_out, base, tv);
    }
    if (alive) {
      a.EX_out[p] = EXn;
      a.EV_out[p] = EVn;
      a.Hflf_out[p] = Hc;
      a.dwell[p] = dwell;
      a.trans[p] = (uint8_t)k;
    }
    stamp_it = 0;
    ROWS_STAMP(15);
  }
  for (int o = 32; o > 0; o >>= 1) first_bad = min(first_bad, __shfl_xor(first_bad, o));
  if (first_bad != 0x7fffffff && lane == 0) {
    a.ctl->failed = 1;
    atomicMax(&a.ctl->inv_iter, 0x7fffffff - (a.iter + first_bad));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_it * 4; i += 64 * W) {
    const unsigned t = (unsigned)(tally[i >> 2][(i >> 1) & 1] >> (32 * (i & 1)));
    if (t) atomicAdd(&a.stats[i], (unsigned long long)t);
  }
}

// particles per workgroup of the jump-process launch = 256 x kDecideSub.  A workgroup ends with FOUR global atomics on
// addresses every workgroup of the launch uses -- its movers' place in the next iteration's list (returning), the l / f / r
// tallies -- and same-address atomics are served one after another, ~12 ns each (tools/microbench/atomic_one_address.hip:
// 3 906 workgroups, one counter + three tallies: 55 us per launch; 977 workgroups: 14 us): with 256 particles per
// workgroup they, not the jump process, were most of C4's 0.074 ms launch.  Measured for C4 (10^6 particles), 1 / 2 / 4 / 8
// sub-blocks per workgroup: 74 / 56 / 60 / 65 us (without R-movers 73 / 48 / 46 / 50: more particles per workgroup also
// means their sub-blocks one after the other, and more movers for its lane groups).
constexpr int kDecideSub = 2;

template <typename T, int E>
struct DecideShared {
  typename VecOf<T>::type stash[4][E / VecOf<T>::n][64];
  int movers[256 * kDecideSub];   // this workgroup's movers: particle of the workgroup << 2 | move
  int n_f, n_r, list_base;
  unsigned tally[3];
  int any_bad;
};
template <typename T, int E>
__device__ __forceinline__ void decide_block(const JumpDecideArgs<T>& a, DecideShared<T, E>& sh, int vblock) {
  auto& stash = sh.stash;
  auto& movers = sh.movers;
  int& n_f = sh.n_f;
  int& n_r = sh.n_r;
  int& list_base = sh.list_base;
  auto& tally = sh.tally;
  int& any_bad = sh.any_bad;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    n_f = 0;
    n_r = 0;
    any_bad = 0;
  }
  if (threadIdx.x < 3) tally[threadIdx.x] = 0;
  __syncthreads();
  // ---- the jump process, one lane per particle ------------------------------------------------------------------------
#pragma unroll 1
  for (int sb = 0; sb < kDecideSub; ++sb) {
    const int in_wg = sb * 256 + (int)threadIdx.x;
    const int64_t p_raw = (int64_t)vblock * (256 * kDecideSub) + in_wg;
    const bool alive = p_raw < a.N;
    const int64_t p = alive ? p_raw : a.N - 1;
    const T EX0 = a.EX_in[p], EV0 = a.EV_in[p];
    const T H0 = EX0 + EV0;                   // HMCState.H (hmc_state.py:80-84)
    const T HL = a.EX_out[p] + a.EV_out[p];
    T Hflf = a.Hflf_in[p];
    if (!(Hflf == Hflf)) Hflf = a.Hwork[p];   // cold cache: integrated by this iteration's mjhmc_traj_kernel
    JumpArgs<T> ja;
    ja.p_r = a.p_r;
    ja.rexp = nullptr;
    ja.runif = nullptr;
    ja.N = a.N;
    LaneMap m1;
    m1.j = 0;
    m1.G = 1;
    m1.D = 1;
    m1.CH = 1;
    m1.lane0 = 0;
    m1.wpp = 0;
    int k = 0;
    double dwell = 0.0;
    bool bad = false;
    decide<T, false>(ja, a.key, m1, H0, HL, Hflf, p, (uint32_t)(a.first_pid + p), k, dwell, bad);
    if (alive) {
      a.Hflf_out[p] = (k == 0) ? H0 : (T)__builtin_nan("");   // L: the pre-move state becomes the cached inverse-L state
      a.dwell[p] = dwell;
      a.dwell_ring[p] = dwell;
      a.trans[p] = (uint8_t)k;
      if (k != 0) {                            // F / R keep the position; an R-mover's kinetic energy follows below
        a.EX_out[p] = EX0;
        a.EV_out[p] = EV0;
      }
      if (bad) any_bad = 1;
    }
    const unsigned long long b0 = __ballot(alive && k == 0), b1 = __ballot(alive && k == 1), b2 = __ballot(alive && k == 2);
    if (lane == 0) {
      if (b0) atomicAdd(&tally[0], (unsigned)__popcll(b0));
      if (b1) atomicAdd(&tally[1], (unsigned)__popcll(b1));
      if (b2) atomicAdd(&tally[2], (unsigned)__popcll(b2));
    }
    // the R-movers from the front of the array, the F-movers from its back: the lane groups that redraw a momentum
    // (Philox + Box-Muller: ~1000 instructions) sit together in as few waves as possible
    const unsigned long long below = (1ull << lane) - 1ull;
    if (b2) {
      int at = 0;
      if (lane == 0) at = atomicAdd(&n_r, (int)__popcll(b2));
      at = __shfl(at, 0);
      if (alive && k == 2) movers[at + (int)__popcll(b2 & below)] = (in_wg << 2) | 2;
    }
    if (b1) {
      int at = 0;
      if (lane == 0) at = atomicAdd(&n_f, (int)__popcll(b1));
      at = __shfl(at, 0);
      if (alive && k == 1) movers[256 * kDecideSub - 1 - (at + (int)__popcll(b1 & below))] = (in_wg << 2) | 1;
    }
  }
  __syncthreads();
  const int nr = n_r, nf = n_f, nm = nr + nf;
  if (threadIdx.x == 0) {
    if (any_bad) {  // draw_from's ValueError (utils.py:43-48): the host rolls this attempt back
      a.ctl->failed_iter = a.iter;
      __threadfence();
      a.ctl->failed = 1;
    }
    list_base = nm ? atomicAdd(a.next_count, nm) : 0;
  }
  if (threadIdx.x < 3 && tally[threadIdx.x]) atomicAdd(&a.stats[threadIdx.x], (unsigned long long)tally[threadIdx.x]);
  __syncthreads();
  auto mover = [&](int i) { return movers[i < nr ? i : 256 * kDecideSub - 1 - (i - nr)]; };
  // every move but L clears the cache (markov_jump_hmc.py:409-410): the movers are the next iteration's list
  for (int i = threadIdx.x; i < nm; i += 256) a.next_list[list_base + i] = (int)((int64_t)vblock * (256 * kDecideSub) + (mover(i) >> 2));
  // ---- the movers' successor states, a lane group per mover ------------------------------------------------------------
  const int G = 1 << a.logG;
  LaneMap m;
  m.j = (int)(threadIdx.x & (G - 1));
  m.G = G;
  m.D = a.D;
  m.CH = a.CH;
  m.lane0 = lane & ~(G - 1);
  m.wpp = 0;
  const int ppb = 256 >> a.logG;
  for (int first = 0; first < nm; first += ppb) {
    const int idx = first + (threadIdx.x >> a.logG);
    const bool live = idx < nm;
    const int code = mover(live ? idx : 0);
    const int64_t p = (int64_t)vblock * (256 * kDecideSub) + (code >> 2);
    const int k = code & 3;
    if (__ballot(live) == 0ull) continue;   // (a wave without a mover has nothing to do)
    T x[E], v[E];
    load_row<T, E>(a.X_in + (size_t)p * a.pitch, m, x);
    load_row<T, E>(a.V_in + (size_t)p * a.pitch, m, v);
    const bool r = live && k == 2;
    if (__ballot(r) != 0ull) {  // HMCState.R (hmc_state.py:121-129)
      stash_put<T, E>(stash[wib], lane, v);
      refresh_stash<T, E, false>(stash[wib], lane, nullptr, a.key, (uint32_t)(a.first_pid + p), m, a.r_keep, a.r_mix);
      T vr[E];
      stash_get<T, E>(stash[wib], lane, vr);
      const T evr = kinetic<T, E>(vr, m);
      if (r) {
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = vr[e];
        if (m.j == 0) a.EV_out[p] = evr;
      }
    }
    if (live && k == 1) {       // HMCState.F
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = -v[e];
    }
    if (live) {
      store_row<T, E>(a.X_out + (size_t)p * a.pitch, m, x);
      store_row<T, E>(a.V_out + (size_t)p * a.pitch, m, v);
    }
  }
}


// One launch = the trajectories of one half of the batch (TrajArgs) BESIDE the jump process of the other half
// (JumpDecideArgs), their workgroups interleaved four to one -- the ratio of their grids -- so that every CU holds both
// kinds at any time; either side may be absent (n_traj or n_decide zero: the first and the last launch of a call).
template <class En, typename T, int E>
__global__ __launch_bounds__(256) void mjhmc_step_kernel(const TrajArgs<T> ta, const JumpDecideArgs<T> da, const En en,
                                                         int n_traj, int n_decide) {
  __shared__ DecideShared<T, E> sh;
  if ((n_traj ? ta.ctl : da.ctl)->failed) return;
  const int b = blockIdx.x;
  const int groups = min(n_decide, n_traj / 4);      // [4 trajectory workgroups, 1 deciding workgroup] x groups, then the rest
  int role, vb;                                      // role 0: trajectories, 1: jump process
  if (b < 5 * groups) {
    const int g = b / 5, r = b % 5;
    role = r == 4;
    vb = role ? g : 4 * g + r;
  } else {
    const int rest = b - 5 * groups, left_t = n_traj - 4 * groups;
    role = rest >= left_t;
    vb = role ? groups + (rest - left_t) : 4 * groups + rest;
  }
  if (role == 0) traj_block<En, T, E>(ta, en, vb);
  else decide_block<T, E>(da, sh, vb);
}

// ------------------------------------------------------------------------------------------
// HMCState.leapfrog / HMCState.L as an operator on caller-supplied states (hmc_state.py:86-100): the reference's
// literal operation order (the EXACT trajectory), then EV, EX and dE/dX of the end point.
// ------------------------------------------------------------------------------------------
template <class En, typename T, int E>
__global__ __launch_bounds__(256) void mjhmc_leap_kernel(const LeapArgs<T> a, const En en) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int G = 1 << a.logG;
  const int64_t p_raw = tid >> a.logG;
  const bool alive = p_raw < a.N;
  const int64_t p = alive ? p_raw : a.N - 1;
  LaneMap m;
  m.j = (int)(tid & (G - 1));
  m.G = G;
  m.D = a.D;
  m.CH = a.CH;
  m.lane0 = (int)((threadIdx.x & 63) & ~(G - 1));
  m.wpp = 0;
  T x[E], v[E];
  load_row<T, E>(a.X + (size_t)p * a.pitch, m, x);
  load_row<T, E>(a.V + (size_t)p * a.pitch, m, v);
  const auto lc = en.template local<E>(m);
  trajectory<En, T, E, true>(en, lc, m, x, v, a.L, a.eps, a.chalf);
  if (alive) {
    store_row<T, E>(a.X_out + (size_t)p * a.pitch, m, x);
    store_row<T, E>(a.V_out + (size_t)p * a.pitch, m, v);
  }
  if (a.G) {
    T g[E];
    const auto ctx = en.prep(x, m);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int d = dim_of<T, E>(m, e);
      g[e] = (d < a.D) ? en.template grad<E>(x[e], e, d, ctx, lc) : T(0);
    }
    if (alive) store_row<T, E>(a.G + (size_t)p * a.pitch, m, g);
  }
  if (a.EX) {
    const T ex = en.energy(x, m, lc);
    if (alive && m.j == 0) a.EX[p] = ex;
  }
  if (a.EV) {
    const T ev = kinetic<T, E>(v, m);
    if (alive && m.j == 0) a.EV[p] = ev;
  }
}

#ifndef __HIPCC_RTC__
template <class En, typename T, int E>
inline void launch_leap_t(const LeapArgs<T>& a, const En& en, hipStream_t st) {
  const int64_t threads = a.N << a.logG;
  hipLaunchKernelGGL((mjhmc_leap_kernel<En, T, E>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, a, en);
}

#endif  // !__HIPCC_RTC__

// ------------------------------------------------------------------------------------------
// evaluation kernel: E(X), dEdX(X), optionally kinetic energy / generated initial momentum
// (HMCState.__init__, hmc_state.py:24-39; Distribution.E/dEdX, distributions.py:62-81)
// ------------------------------------------------------------------------------------------

template <class En, typename T, int E>
__global__ __launch_bounds__(256) void mjhmc_eval_kernel(const EvalArgs<T> a, const En en) {
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int G = 1 << a.logG;
  const int64_t p_raw = tid >> a.logG;
  const bool alive = p_raw < a.N;
  const int64_t p = alive ? p_raw : a.N - 1;
  LaneMap m;
  m.j = (int)(tid & (G - 1));
  m.G = G;
  m.D = a.D;
  m.CH = a.CH;
  m.lane0 = (int)((threadIdx.x & 63) & ~(G - 1));
  m.wpp = 0;
  T x[E];
  load_row<T, E>(a.X + (size_t)p * a.pitch, m, x);
  const auto lc = en.template local<E>(m);
  if (a.G) {
    T g[E];
    const auto ctx = en.prep(x, m);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int d = dim_of<T, E>(m, e);
      g[e] = (d < a.D) ? en.template grad<E>(x[e], e, d, ctx, lc) : T(0);
    }
    if (alive) store_row<T, E>(a.G + (size_t)p * a.pitch, m, g);
  }
  if (a.E) {
    const T ex = en.energy(x, m, lc);
    if (alive && m.j == 0) a.E[p] = ex;
  }
  if (a.EV) {
    T v[E];
    if (a.V_out) {
      refresh_noise<T, E>(nullptr, a.key, (uint32_t)(a.first_pid + p), m, v);
      if (alive) store_row<T, E>(a.V_out + (size_t)p * a.pitch, m, v);
    } else {
      load_row<T, E>(a.V + (size_t)p * a.pitch, m, v);
    }
    const T ev = kinetic<T, E>(v, m);
    if (alive && m.j == 0) a.EV[p] = ev;
  }
}

// ------------------------------------------------------------------------------------------
// launch plumbing shared by the per-energy translation units
// ------------------------------------------------------------------------------------------

struct LaunchShape {
  int E;      // elements per lane (template arg)
  int logG;
  int pitch;
  int CH;
};

// host-visible parameter block; each energy TU turns it into its functor
constexpr int kParamPad = 4096;  // device parameter vectors are zero padded to this many elements

struct EnergyParams {
  int kind;
  int ndims;
  double p[8];           // scalar params
  const void* dev_f64;   // device arrays (already in the state dtype), zero padded to pitch
  const void* dev_f32;
};

#ifndef __HIPCC_RTC__
// Persistent launch: as many 256-thread blocks as the device keeps resident for this kernel
// (occupancy query, cached per instantiation), never more than there are slots to hand out.
template <class En, typename T, int E, int MODE, bool REPLAY, bool FULLROW, int WPP = 0, bool FUSED = false>
inline void launch_jump_r(const JumpArgs<T>& a, const En& en, hipStream_t st) {
  static int resident_blocks = 0;
  if (resident_blocks == 0) {
    int dev = 0, per_cu = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mjhmc_jump_kernel<En, T, E, MODE, REPLAY, FULLROW, WPP, FUSED>,
                                                       256, 0);
    resident_blocks = std::max(1, per_cu) * std::max(1, cus);
  }
  const int64_t nslots = a.Npad >> (6 - a.logG);
  const int64_t want = (nslots + 3) / 4;
  const unsigned grid = (unsigned)std::min<int64_t>(want, resident_blocks);
  hipLaunchKernelGGL((mjhmc_jump_kernel<En, T, E, MODE, REPLAY, FULLROW, WPP, FUSED>), dim3(grid), dim3(256), 0, st, a, en);
}

// fused MarkovJumpHMC launches of the energies with a row form run a lane per particle for rows held by 2 or 4 lanes elsewhere
inline bool fused_rows_shape(int mode, int logG, int ab) {
  return mode == kModeMJHMC && (logG == 1 || logG == 2) && !(ab & kAbNoRows);
}
// persistent: as many one-wave workgroups as the device keeps resident (one per SIMD), never more than there are tiles.
// The product's form is the relay kernel (four-wave workgroups, one per CU, the workgroup's cold caches pooled); the
// one-wave form stays as the test build's A/B partner (kAbNoRelay) -- the two must agree bit for bit.
template <class En, typename T, int E, int LOGG, bool FULL>
inline void launch_fused_rows(const JumpArgs<T>& a, const En& en, hipStream_t st) {
  static int resident_blocks = 0, resident_relay = 0;
  if (resident_blocks == 0) {
    int dev = 0, per_cu = 0, per_cu_relay = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mjhmc_fused_rows_kernel<En, T, E, LOGG, FULL>, 64, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_relay, mjhmc_fused_rows_relay_kernel<En, T, E, LOGG, FULL>,
                                                       64 * kRelayWaves, 0);
    resident_blocks = std::max(1, per_cu) * std::max(1, cus);
    resident_relay = std::max(1, per_cu_relay) * std::max(1, cus);
  }
  const int64_t tiles = a.Npad >> 6;
  // The relay trades a trajectory per wave and iteration for a quarter of one + four hand-overs through LDS: it wins from
  // ~12 leapfrog steps up (tools/sweep_L.py 32 1000000 funnel, profiles/r06/sweep_L_c4.txt: relay 0.110 + 0.0045 L ms per
  // iteration, one-wave kernel 0.071 + 0.0078 L); shorter trajectories keep the one-wave kernel.  The same bits either way.
  const bool relay = (a.ab & kAbForceRelay) || (!(a.ab & kAbNoRelay) && a.L >= RelayMinL<En>::value);
  if (relay) {
    const unsigned grid = (unsigned)std::min<int64_t>((tiles + kRelayWaves - 1) / kRelayWaves, resident_relay);
    hipLaunchKernelGGL((mjhmc_fused_rows_relay_kernel<En, T, E, LOGG, FULL>), dim3(grid), dim3(64 * kRelayWaves), 0, st, a, en);
    return;
  }
  const unsigned grid = (unsigned)std::min<int64_t>(tiles, resident_blocks);
  hipLaunchKernelGGL((mjhmc_fused_rows_kernel<En, T, E, LOGG, FULL>), dim3(grid), dim3(64), 0, st, a, en);
}

// Replay needs every recorded stream of the mode; it is a test path and exists only in the
// predicated (non-FULLROW) form, as do the control-arm samplers.
template <class En, typename T, int E>
inline void launch_jump_t(const JumpArgs<T>& a, const En& en, hipStream_t st) {
  const bool full = a.CH == (E / VecOf<T>::n) << a.logG;
  const bool replay = a.noise != nullptr;
  if constexpr (En::kFuse) if (a.n_fuse > 0) {  // several iterations per launch (counter RNG only)
    if constexpr (HasRowForm<En>::value && sizeof(T) == 8 && E == 8) {
      if (fused_rows_shape(a.mode, a.logG, a.ab)) {   // a lane per particle (mjhmc_fused_rows_kernel)
        const bool full_row = a.CH == 4 << a.logG;
        if (a.logG == 2 && full_row) launch_fused_rows<En, T, E, 2, true>(a, en, st);
        else if (a.logG == 2) launch_fused_rows<En, T, E, 2, false>(a, en, st);
        else if (full_row) launch_fused_rows<En, T, E, 1, true>(a, en, st);
        else launch_fused_rows<En, T, E, 1, false>(a, en, st);
        return;
      }
    }
    if (a.mode == kModeMJHMC) {
      if (full && a.logG == 6 && !(a.ab & kAbNoBlockDecide)) launch_jump_r<En, T, E, kModeMJHMC, false, true, 5, true>(a, en, st);
      else if (full && a.logG == 5 && !(a.ab & kAbNoBlockDecide)) launch_jump_r<En, T, E, kModeMJHMC, false, true, 6, true>(a, en, st);
      else if (full && a.logG == 6) launch_jump_r<En, T, E, kModeMJHMC, false, true, 1, true>(a, en, st);
      else if (full && a.logG == 2 && !(a.ab & kAbNoQuad)) launch_jump_r<En, T, E, kModeMJHMC, false, true, 3, true>(a, en, st);
      else if (full) launch_jump_r<En, T, E, kModeMJHMC, false, true, 0, true>(a, en, st);
      else launch_jump_r<En, T, E, kModeMJHMC, false, false, 0, true>(a, en, st);
    } else if (a.mode == kModeCT) {
      launch_jump_r<En, T, E, kModeCT, false, false, 0, true>(a, en, st);
    } else {
      launch_jump_r<En, T, E, kModeControl, false, false, 0, true>(a, en, st);
    }
    return;
  }
  if (a.mode == kModeMJHMC) {
    if (replay) launch_jump_r<En, T, E, kModeMJHMC, true, false>(a, en, st);
    else if (full && a.logG == 6 && !(a.ab & kAbNoWpp)) launch_jump_r<En, T, E, kModeMJHMC, false, true, 1>(a, en, st);
    else if (full && a.logG == 2 && !(a.ab & kAbNoQuad)) launch_jump_r<En, T, E, kModeMJHMC, false, true, 3>(a, en, st);
    else if (full) launch_jump_r<En, T, E, kModeMJHMC, false, true>(a, en, st);
    else launch_jump_r<En, T, E, kModeMJHMC, false, false>(a, en, st);
  } else if (a.mode == kModeCT) {
    if (replay) launch_jump_r<En, T, E, kModeCT, true, false>(a, en, st);
    else launch_jump_r<En, T, E, kModeCT, false, false>(a, en, st);
  } else {
    if (replay) launch_jump_r<En, T, E, kModeControl, true, false>(a, en, st);
    else launch_jump_r<En, T, E, kModeControl, false, false>(a, en, st);
  }
}

// trajectories: one workgroup per 256 >> logG particles + the list's walkers in front; jump process: one per 256 x kDecideSub particles
template <class En, typename T, int E>
inline void launch_step_t(const TrajArgs<T>* ta, const JumpDecideArgs<T>* da, const En& en, hipStream_t st) {
  int n_traj = 0, n_decide = 0;
  if (ta) {
    const int64_t ppb = 256 >> ta->logG;
    n_traj = (int)((ta->N + ppb - 1) / ppb) + ta->inv_blocks;
  }
  if (da) n_decide = (int)((da->N + 256 * kDecideSub - 1) / (256 * kDecideSub));
  if constexpr (HasRowForm<En>::value && sizeof(T) == 8 && E == 8) {
    if (ta && !da && ta->rows && (ta->logG == 1 || ta->logG == 2)) {   // a lane per particle, a wavefront per workgroup
      const int64_t fwd = (ta->N + 63) / 64;
      const int n_walk = (int)std::max<int64_t>(1, std::min<int64_t>(fwd / 8, 4096));
      const dim3 grid((unsigned)(fwd + n_walk));
      const bool full = ta->CH == 4 << ta->logG;
      if (ta->logG == 2 && full) hipLaunchKernelGGL((mjhmc_traj_rows_kernel<En, T, E, 2, true>), grid, dim3(64), 0, st, *ta, en, n_walk);
      else if (ta->logG == 2) hipLaunchKernelGGL((mjhmc_traj_rows_kernel<En, T, E, 2, false>), grid, dim3(64), 0, st, *ta, en, n_walk);
      else if (full) hipLaunchKernelGGL((mjhmc_traj_rows_kernel<En, T, E, 1, true>), grid, dim3(64), 0, st, *ta, en, n_walk);
      else hipLaunchKernelGGL((mjhmc_traj_rows_kernel<En, T, E, 1, false>), grid, dim3(64), 0, st, *ta, en, n_walk);
      return;
    }
  }
  const TrajArgs<T> t = ta ? *ta : TrajArgs<T>{};
  const JumpDecideArgs<T> d = da ? *da : JumpDecideArgs<T>{};
  hipLaunchKernelGGL((mjhmc_step_kernel<En, T, E>), dim3((unsigned)(n_traj + n_decide)), dim3(256), 0, st, t, d, en, n_traj,
                     n_decide);
}

template <class En, typename T, int E>
inline void launch_eval_t(const EvalArgs<T>& a, const En& en, hipStream_t st) {
  const int64_t threads = a.N << a.logG;
  const unsigned grid = (unsigned)((threads + 255) / 256);
  hipLaunchKernelGGL((mjhmc_eval_kernel<En, T, E>), dim3(grid), dim3(256), 0, st, a, en);
}

// E choices: f64 {2, 8, 16}, f32 {4, 16, 32}  (1, 4, 8 chunks per lane)
#define MJHMC_DEFINE_ENERGY_LAUNCHERS(NAME, MAKE64, MAKE32)                                               \
  void NAME##_jump_f64(const JumpArgs<double>& a, const EnergyParams& ep, int E, hipStream_t st) {        \
    const auto en = MAKE64(ep);                                                                           \
    if (E == 2) launch_jump_t<decltype(en), double, 2>(a, en, st);                                        \
    else if (E == 8) launch_jump_t<decltype(en), double, 8>(a, en, st);                                   \
    else launch_jump_t<decltype(en), double, 16>(a, en, st);                                              \
  }                                                                                                       \
  void NAME##_jump_f32(const JumpArgs<float>& a, const EnergyParams& ep, int E, hipStream_t st) {         \
    const auto en = MAKE32(ep);                                                                           \
    if (E == 4) launch_jump_t<decltype(en), float, 4>(a, en, st);                                         \
    else if (E == 16) launch_jump_t<decltype(en), float, 16>(a, en, st);                                  \
    else launch_jump_t<decltype(en), float, 32>(a, en, st);                                               \
  }                                                                                                       \
  void NAME##_leap_f64(const LeapArgs<double>& a, const EnergyParams& ep, int E, hipStream_t st) {        \
    const auto en = MAKE64(ep);                                                                           \
    if (E == 2) launch_leap_t<decltype(en), double, 2>(a, en, st);                                        \
    else if (E == 8) launch_leap_t<decltype(en), double, 8>(a, en, st);                                   \
    else launch_leap_t<decltype(en), double, 16>(a, en, st);                                              \
  }                                                                                                       \
  void NAME##_leap_f32(const LeapArgs<float>& a, const EnergyParams& ep, int E, hipStream_t st) {         \
    const auto en = MAKE32(ep);                                                                           \
    if (E == 4) launch_leap_t<decltype(en), float, 4>(a, en, st);                                         \
    else if (E == 16) launch_leap_t<decltype(en), float, 16>(a, en, st);                                  \
    else launch_leap_t<decltype(en), float, 32>(a, en, st);                                               \
  }                                                                                                       \
  void NAME##_step_f64(const TrajArgs<double>* ta, const JumpDecideArgs<double>* da, const EnergyParams& ep, int E, hipStream_t st) { \
    const auto en = MAKE64(ep);                                                                           \
    if (E == 2) launch_step_t<decltype(en), double, 2>(ta, da, en, st);                                   \
    else if (E == 8) launch_step_t<decltype(en), double, 8>(ta, da, en, st);                              \
    else launch_step_t<decltype(en), double, 16>(ta, da, en, st);                                         \
  }                                                                                                       \
  void NAME##_step_f32(const TrajArgs<float>* ta, const JumpDecideArgs<float>* da, const EnergyParams& ep, int E, hipStream_t st) { \
    const auto en = MAKE32(ep);                                                                           \
    if (E == 4) launch_step_t<decltype(en), float, 4>(ta, da, en, st);                                    \
    else if (E == 16) launch_step_t<decltype(en), float, 16>(ta, da, en, st);                             \
    else launch_step_t<decltype(en), float, 32>(ta, da, en, st);                                          \
  }                                                                                                       \
  void NAME##_eval_f64(const EvalArgs<double>& a, const EnergyParams& ep, int E, hipStream_t st) {        \
    const auto en = MAKE64(ep);                                                                           \
    if (E == 2) launch_eval_t<decltype(en), double, 2>(a, en, st);                                        \
    else if (E == 8) launch_eval_t<decltype(en), double, 8>(a, en, st);                                   \
    else launch_eval_t<decltype(en), double, 16>(a, en, st);                                              \
  }                                                                                                       \
  void NAME##_eval_f32(const EvalArgs<float>& a, const EnergyParams& ep, int E, hipStream_t st) {         \
    const auto en = MAKE32(ep);                                                                           \
    if (E == 4) launch_eval_t<decltype(en), float, 4>(a, en, st);                                         \
    else if (E == 16) launch_eval_t<decltype(en), float, 16>(a, en, st);                                  \
    else launch_eval_t<decltype(en), float, 32>(a, en, st);                                               \
  }

#define MJHMC_DECLARE_ENERGY_LAUNCHERS(NAME)                                                       \
  void NAME##_jump_f64(const JumpArgs<double>&, const EnergyParams&, int, hipStream_t);           \
  void NAME##_jump_f32(const JumpArgs<float>&, const EnergyParams&, int, hipStream_t);            \
  void NAME##_leap_f64(const LeapArgs<double>&, const EnergyParams&, int, hipStream_t);           \
  void NAME##_leap_f32(const LeapArgs<float>&, const EnergyParams&, int, hipStream_t);            \
  void NAME##_step_f64(const TrajArgs<double>*, const JumpDecideArgs<double>*, const EnergyParams&, int, hipStream_t); \
  void NAME##_step_f32(const TrajArgs<float>*, const JumpDecideArgs<float>*, const EnergyParams&, int, hipStream_t);  \
  void NAME##_eval_f64(const EvalArgs<double>&, const EnergyParams&, int, hipStream_t);           \
  void NAME##_eval_f32(const EvalArgs<float>&, const EnergyParams&, int, hipStream_t);

MJHMC_DECLARE_ENERGY_LAUNCHERS(iso)
MJHMC_DECLARE_ENERGY_LAUNCHERS(diag)
MJHMC_DECLARE_ENERGY_LAUNCHERS(rough)
MJHMC_DECLARE_ENERGY_LAUNCHERS(mm)
MJHMC_DECLARE_ENERGY_LAUNCHERS(funnel_neal)
MJHMC_DECLARE_ENERGY_LAUNCHERS(funnel_ref)

#endif  // !__HIPCC_RTC__

}  // namespace mjhmc
