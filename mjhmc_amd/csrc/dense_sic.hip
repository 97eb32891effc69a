// SparseImageCode (sparse-coding posterior over coefficients, mjhmc/misc/tf_distributions.py:204-272) on
// the bf16 matrix cores: bf16 state in HBM, bf16 MFMA operands, fp32 accumulation and fp32 integrator
// registers (BASELINE.json configs[4]).  Built for one patch per particle (n_patches = 1), n_coeffs = 1024,
// img_size = 256:
//
//   resid = B a - y ,  E = 1/2 |resid|^2 + lambda * sum log(1 + a^2)            (Cauchy prior, :262-267)
//   dE/da = B^T resid + lambda * 2a / (1 + a^2)                                 (or lambda * sign(a), Laplace)
//
// One workgroup = 8 waves (two per SIMD) owns a tile of 32 particles.  Wave w holds coefficient rows
// [128w, 128w+128) of X and V as fp32 MFMA accumulator tiles (4 blocks of 32 rows) and image rows
// [32w, 32w+32) of the residual (1 block).  v_mfma_f32_32x32x16_bf16 throughout:
//   GEMM1  resid[i][n] = sum_c B[i][c] a[c][n] - y[i]     64 k-steps x 1 block  per wave
//   GEMM2  V[c][n]    += sum_i B[i][c] (s * resid[i][n])  16 k-steps x 4 blocks per wave
// GEMM2 accumulates straight into the momentum registers: the kick "V += s * dE/dX" is the MFMA's C
// operand, with the step scale s = -eps/2 (first/last half kick) or -eps (the two half kicks between
// drifts, merged) folded into the bf16 residual when it is published.  Accumulator registers 8t..8t+7
// converted to bf16 are directly the B fragment of k-step t (rows 16t + 8(j>>2) + 4h + (j&3)); the
// dictionary is pre-permuted on the host into that k-order ("A1", "A2"), 16 contiguous bytes per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "dense_sic.hpp"

namespace mjhmc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

constexpr int kP = 32;
constexpr int kC = kSicCoeffs;  // 1024
constexpr int kI = kSicImg;     // 256

__device__ __forceinline__ int acc_row(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

struct CTile {
  f32x16 b[4];  // this wave's 128 coefficient rows x 32 particles
};

// bf16 state rows [*][1024]: lane (c, h) reads its 16 groups of 4 consecutive coefficients (8 bytes each)
__device__ __forceinline__ void ctile_load(const __bf16* base, int64_t p, int w, int h, CTile& t) {
  const __bf16* row = base + (size_t)p * kC + 128 * w + 4 * h;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bf16x4 v = *reinterpret_cast<const bf16x4*>(row + 32 * b + 8 * g);
#pragma unroll
      for (int k = 0; k < 4; ++k) t.b[b][4 * g + k] = (float)v[k];
    }
}

__device__ __forceinline__ void ctile_store(__bf16* base, int64_t p, int w, int h, const CTile& t) {
  __bf16* row = base + (size_t)p * kC + 128 * w + 4 * h;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (__bf16)t.b[b][4 * g + k];
      *reinterpret_cast<bf16x4*>(row + 32 * b + 8 * g) = v;
    }
}

struct SicShared {
  f32x4 pubA[8][4][2][64];  // 64 KB: a as B fragments, [wave][block][k-step half][lane]
  f32x4 pubR[8][2][64];     // 16 KB: scaled residual as B fragments
  float red[2][8][kP];
  int move[kP];
};

__device__ __forceinline__ f32x4 frag_of(const f32x16& acc, int s, float scale) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (__bf16)(acc[8 * s + j] * scale);
  return __builtin_bit_cast(f32x4, f);
}

// residual of the tile at the X held in x (GEMM1).  Leaves it in `res` (fp32 accumulator layout).
__device__ __forceinline__ void sic_residual(const SicModel& mdl, SicShared& sh, int w, int c, int h, int lane,
                                             const CTile& x, f32x16& res) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    sh.pubA[w][b][0][lane] = frag_of(x.b[b], 0, 1.0f);
    sh.pubA[w][b][1][lane] = frag_of(x.b[b], 1, 1.0f);
  }
  __syncthreads();
  {  // res starts at -y[i]
    const float* yv = mdl.y + 32 * w + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(yv + 8 * g);
#pragma unroll
      for (int k = 0; k < 4; ++k) res[4 * g + k] = -v[k];
    }
  }
  // A1[kstep][h][i][8 bf16]: 16 bytes per lane, lanes of a half contiguous
  const f32x4* a1 = reinterpret_cast<const f32x4*>(mdl.A1) + (size_t)h * kI + 32 * w + c;
  // 16 chunks of 4 k-steps, two per trip; chunk n+1's A fragments are in flight while chunk n's MFMAs run
  f32x4 fa[2][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) fa[0][t] = a1[(size_t)t * 2 * kI];
#pragma unroll 1
  for (int ch = 0; ch < 16; ch += 2) {
#pragma unroll
    for (int t = 0; t < 4; ++t) fa[1][t] = a1[(size_t)((ch + 1) * 4 + t) * 2 * kI];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ks = ch * 4 + t;
      const f32x4 fb = sh.pubA[ks >> 3][(ks >> 1) & 3][ks & 1][lane];
      res = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[0][t]), __builtin_bit_cast(bf16x8, fb),
                                                    res, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int nxt = ch + 2 < 16 ? ch + 2 : 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) fa[0][t] = a1[(size_t)(nxt * 4 + t) * 2 * kI];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ks = (ch + 1) * 4 + t;
      const f32x4 fb = sh.pubA[ks >> 3][(ks >> 1) & 3][ks & 1][lane];
      res = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[1][t]), __builtin_bit_cast(bf16x8, fb),
                                                    res, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// acc[c][n] += sum_i B[i][c] * (scale * res[i][n]) + scale * prior'(x)   (GEMM2 into the caller's tile)
template <bool CAUCHY>
__device__ __forceinline__ void sic_kick(const SicModel& mdl, SicShared& sh, int w, int c, int h, int lane,
                                         const f32x16& res, const CTile& x, float scale, CTile& acc) {
  sh.pubR[w][0][lane] = frag_of(res, 0, scale);
  sh.pubR[w][1][lane] = frag_of(res, 1, scale);
  __syncthreads();
  const f32x4* a2 = reinterpret_cast<const f32x4*>(mdl.A2) + (size_t)h * kC + 128 * w + c;
#pragma unroll 1
  for (int ks = 0; ks < 16; ks += 2) {
    f32x4 fa[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int b = 0; b < 4; ++b) fa[u][b] = a2[(size_t)(ks + u) * 2 * kC + 32 * b];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const f32x4 fb = sh.pubR[(ks + u) >> 1][(ks + u) & 1][lane];
#pragma unroll
      for (int b = 0; b < 4; ++b)
        acc.b[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[u][b]),
                                                           __builtin_bit_cast(bf16x8, fb), acc.b[b], 0, 0, 0);
    }
  }
  const float sl = scale * mdl.lambda;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float a = x.b[b][q];
      if (CAUCHY) acc.b[b][q] += sl * (2.0f * a / (1.0f + a * a));
      else acc.b[b][q] += sl * (a > 0.f ? 1.0f : (a < 0.f ? -1.0f : 0.0f));
    }
}

__device__ __forceinline__ float half_swap_sum(float s) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(s), __float_as_int(s), false, false);
  return __int_as_float(sw[0]) + __int_as_float(sw[1]);
}

// E(x) per particle from the residual at x:  1/2 |res|^2 + lambda * prior(x)
template <bool CAUCHY>
__device__ __forceinline__ float sic_energy(const SicModel& mdl, SicShared& sh, int w, int c, int h, const f32x16& res,
                                            const CTile& x) {
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += 0.5f * res[q] * res[q];
  float pr = 0.f;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float a = x.b[b][q];
      pr += CAUCHY ? logf(1.0f + a * a) : fabsf(a);
    }
  const float part = half_swap_sum(s + mdl.lambda * pr);
  if (h == 0) sh.red[0][w][c] = part;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) tot += sh.red[0][k][c];
  __syncthreads();
  return tot;
}

__device__ __forceinline__ float sic_kinetic(SicShared& sh, int w, int c, int h, const CTile& v) {
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int q = 0; q < 16; ++q) s += v.b[b][q] * v.b[b][q];
  const float part = half_swap_sum(s);
  if (h == 0) sh.red[1][w][c] = part;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) tot += sh.red[1][k][c];
  __syncthreads();
  return tot / 2.0f;
}

// L leapfrog steps with the half kicks between drifts merged (bf16 operands make the reference's
// separate roundings meaningless).  Returns E(x_new); x, v updated in place.
template <bool CAUCHY>
__device__ __forceinline__ float sic_trajectory(const SicModel& mdl, SicShared& sh, int w, int c, int h, int lane,
                                                CTile& x, CTile& v, int L, float eps, float chalf) {
  f32x16 res;
  sic_residual(mdl, sh, w, c, h, lane, x, res);
  sic_kick<CAUCHY>(mdl, sh, w, c, h, lane, res, x, chalf, v);
  for (int s = 1; s <= L; ++s) {
#pragma unroll
    for (int b = 0; b < 4; ++b) x.b[b] = x.b[b] + eps * v.b[b];
    sic_residual(mdl, sh, w, c, h, lane, x, res);
    sic_kick<CAUCHY>(mdl, sh, w, c, h, lane, res, x, s < L ? 2.0f * chalf : chalf, v);
  }
  // the successor position is stored in bf16 (and GEMM1 already saw bf16(x)): evaluate the prior on
  // what will be stored, so EX is the energy of the stored state
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int q = 0; q < 16; ++q) x.b[b][q] = (float)(__bf16)x.b[b][q];
  return sic_energy<CAUCHY>(mdl, sh, w, c, h, res, x);
}

// ---------------------------------------------------------------------------------------------------
template <bool CAUCHY>
__global__ __launch_bounds__(512, 2) void sic_eval_kernel(const SicEvalArgs a, const SicModel mdl) {
  __shared__ SicShared sh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t p = tile * kP + c;
    CTile x;
    ctile_load(a.X, p, w, h, x);
    f32x16 res;
    sic_residual(mdl, sh, w, c, h, lane, x, res);
    if (a.G) {
      CTile g;
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) g.b[b][q] = 0.f;
      sic_kick<CAUCHY>(mdl, sh, w, c, h, lane, res, x, 1.0f, g);
      float* row = a.G + (size_t)p * kC + 128 * w + 4 * h;  // dE/dX is handed out in float32
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          f32x4 o;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = g.b[b][4 * gq + k];
          *reinterpret_cast<f32x4*>(row + 32 * b + 8 * gq) = o;
        }
    }
    const float ex = sic_energy<CAUCHY>(mdl, sh, w, c, h, res, x);
    if (a.E && w == 0 && h == 0) a.E[p] = ex;
    if (a.EV) {
      CTile v;
      if (a.V_gen) {
        const uint32_t pid = (uint32_t)(a.first_pid + (p < a.N ? p : 0));
#pragma unroll 1
        for (int gq = 0; gq < 16; ++gq) {
          const int b = gq >> 2, g4 = gq & 3;
          const int d = 128 * w + 32 * b + 8 * g4 + 4 * h;
          float z0, z1, z2, z3;
          normal_pair_f32(a.key, pid, (uint32_t)(d >> 1), z0, z1);
          normal_pair_f32(a.key, pid, (uint32_t)((d >> 1) + 1), z2, z3);
          // round to the state dtype so EV matches the stored momentum
          v.b[b][4 * g4 + 0] = (float)(__bf16)z0;
          v.b[b][4 * g4 + 1] = (float)(__bf16)z1;
          v.b[b][4 * g4 + 2] = (float)(__bf16)z2;
          v.b[b][4 * g4 + 3] = (float)(__bf16)z3;
        }
        ctile_store(a.V_gen, p, w, h, v);
      } else {
        ctile_load(a.V, p, w, h, v);
      }
      const float ev = sic_kinetic(sh, w, c, h, v);
      if (w == 0 && h == 0) a.EV[p] = ev;
    }
  }
}

__global__ void sic_cold_list_kernel(const float* __restrict__ Hflf_in, float* __restrict__ Hwork, int64_t N,
                                     int64_t Npad, int* __restrict__ list, int* __restrict__ count, const Control* ctl,
                                     unsigned long long* stats) {
  if (ctl->failed) return;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= Npad) return;
  const float hc = Hflf_in[p];
  Hwork[p] = hc;
  const bool cold = (p < N) && !(hc == hc);
  const unsigned long long m = __ballot(cold);
  if (m == 0ull) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0) {
    base = atomicAdd(count, (int)__popcll(m));
    atomicAdd(&stats[3], (unsigned long long)__popcll(m));
  }
  base = __shfl(base, 0);
  if (cold) list[base + (int)__popcll(m & ((1ull << lane) - 1ull))] = (int)p;
}

template <bool CAUCHY>
__global__ __launch_bounds__(512, 2) void sic_flf_kernel(const SicJumpArgs a, const SicModel mdl) {
  __shared__ SicShared sh;
  if (a.ctl->failed) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  const int ncold = *a.cold_count;
  for (int tile = blockIdx.x; tile * kP < ncold; tile += gridDim.x) {
    const int slot = tile * kP + c;
    const int64_t p = a.cold_list[slot < ncold ? slot : ncold - 1];
    CTile x, v;
    ctile_load(a.X_in, p, w, h, x);
    ctile_load(a.V_in, p, w, h, v);
#pragma unroll
    for (int b = 0; b < 4; ++b) v.b[b] = -v.b[b];
    const float ex = sic_trajectory<CAUCHY>(mdl, sh, w, c, h, lane, x, v, a.L, a.eps, a.chalf);
    const float ev = sic_kinetic(sh, w, c, h, v);
    if (w == 0 && h == 0) a.Hwork[p] = ex + ev;
    __syncthreads();
  }
}

template <bool CAUCHY, bool REPLAY>
__global__ __launch_bounds__(512, 2) void sic_jump_kernel(const SicJumpArgs a, const SicModel mdl) {
  __shared__ SicShared sh;
  if (a.ctl->failed) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  unsigned nL = 0, nF = 0, nR = 0;
  bool any_bad = false;
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t p = tile * kP + c;
    const bool alive = p < a.N;
    const float EX0 = a.EX_in[p], EV0 = a.EV_in[p];
    const float H0 = EX0 + EV0;
    const float Hflf = a.Hwork[p];
    CTile x, v;
    ctile_load(a.X_in, p, w, h, x);
    ctile_load(a.V_in, p, w, h, v);
    const float EXL = sic_trajectory<CAUCHY>(mdl, sh, w, c, h, lane, x, v, a.L, a.eps, a.chalf);
    // the successor state is stored in bf16: report the kinetic energy of what is stored
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) v.b[b][q] = (float)(__bf16)v.b[b][q];
    const float EVL = sic_kinetic(sh, w, c, h, v);
    const float HL = EXL + EVL;
    if (w == 0 && h == 0) {
      const uint32_t pid = (uint32_t)(a.first_pid + (alive ? p : 0));
      double best;
      bool bad;
      const int k = dense_decide<REPLAY>(H0, HL, Hflf, a.p_r, pid, alive ? p : 0, a.N, a.rexp, a.key, best, bad);
      any_bad |= (bad && alive);
      sh.move[c] = k;
      a.dwell[p] = best;
      a.dwell_ring[p] = best;
      a.trans[p] = (uint8_t)k;
      if (alive) {
        nL += (k == 0);
        nF += (k == 1);
        nR += (k == 2);
      }
      a.EX_out[p] = (k == 0) ? EXL : EX0;
      a.EV_out[p] = (k == 0) ? EVL : EV0;
      a.Hflf_out[p] = (k == 0) ? H0 : __builtin_nanf("");
    }
    __syncthreads();
    const int k = sh.move[c];
    const bool tile_has_r = __ballot(k == 2) != 0ull;
    if (k != 0) {  // F / R keep the position
      ctile_load(a.X_in, p, w, h, x);
      ctile_load(a.V_in, p, w, h, v);
      if (k == 1) {
#pragma unroll
        for (int b = 0; b < 4; ++b) v.b[b] = -v.b[b];
      } else {  // HMCState.R (hmc_state.py:121-129), group by group to keep the register budget
        const uint32_t pid = (uint32_t)(a.first_pid + (alive ? p : 0));
#pragma unroll
        for (int b = 0; b < 4; ++b) v.b[b] = v.b[b] * a.r_keep;
        if constexpr (REPLAY) {
          const __bf16* zrow = a.noise + (size_t)(alive ? p : 0) * kC + 128 * w + 4 * h;
#pragma unroll
          for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const bf16x4 z = *reinterpret_cast<const bf16x4*>(zrow + 32 * b + 8 * g4);
#pragma unroll
              for (int kk = 0; kk < 4; ++kk) v.b[b][4 * g4 + kk] += (float)z[kk] * a.r_mix;
            }
        } else {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
#pragma unroll 1
            for (int g4 = 0; g4 < 4; ++g4) {
              const int d = 128 * w + 32 * b + 8 * g4 + 4 * h;
              float z0, z1, z2, z3;
              normal_pair_f32(a.key, pid, (uint32_t)(d >> 1), z0, z1);
              normal_pair_f32(a.key, pid, (uint32_t)((d >> 1) + 1), z2, z3);
              f32x4 add;
              add[0] = z0 * a.r_mix;
              add[1] = z1 * a.r_mix;
              add[2] = z2 * a.r_mix;
              add[3] = z3 * a.r_mix;
              // g4 is a runtime index here: address the four registers through selects
#pragma unroll
              for (int q = 0; q < 16; ++q)
                if ((q >> 2) == g4) v.b[b][q] += add[q & 3];
            }
          }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int q = 0; q < 16; ++q) v.b[b][q] = (float)(__bf16)v.b[b][q];
      }
    }
    if (tile_has_r) {
      const float evr = sic_kinetic(sh, w, c, h, v);
      if (k == 2 && w == 0 && h == 0) a.EV_out[p] = evr;
    }
    ctile_store(a.X_out, p, w, h, x);
    ctile_store(a.V_out, p, w, h, v);
    __syncthreads();
  }
  if (any_bad) {
    a.ctl->failed = 1;
    a.ctl->failed_iter = a.iter;
  }
  __shared__ unsigned tally[3];
  if (threadIdx.x < 3) tally[threadIdx.x] = 0;
  __syncthreads();
  if (nL) atomicAdd(&tally[0], nL);
  if (nF) atomicAdd(&tally[1], nF);
  if (nR) atomicAdd(&tally[2], nR);
  __syncthreads();
  if (threadIdx.x < 3 && tally[threadIdx.x]) atomicAdd(&a.stats[threadIdx.x], (unsigned long long)tally[threadIdx.x]);
}

static int sic_cus() {
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return std::max(1, cus);
}

void sic_launch_jump(const SicJumpArgs& a, const SicModel& mdl, hipStream_t st) {
  (void)hipMemsetAsync(a.cold_count, 0, sizeof(int), st);
  hipLaunchKernelGGL(sic_cold_list_kernel, dim3((unsigned)((a.Npad + 255) / 256)), dim3(256), 0, st, a.Hflf_in, a.Hwork,
                     a.N, a.Npad, a.cold_list, a.cold_count, (const Control*)a.ctl, a.stats);
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, sic_cus());
  const bool replay = a.rexp && a.noise;
  if (mdl.cauchy) {
    hipLaunchKernelGGL(sic_flf_kernel<true>, dim3(grid), dim3(512), 0, st, a, mdl);
    if (replay) hipLaunchKernelGGL((sic_jump_kernel<true, true>), dim3(grid), dim3(512), 0, st, a, mdl);
    else hipLaunchKernelGGL((sic_jump_kernel<true, false>), dim3(grid), dim3(512), 0, st, a, mdl);
  } else {
    hipLaunchKernelGGL(sic_flf_kernel<false>, dim3(grid), dim3(512), 0, st, a, mdl);
    if (replay) hipLaunchKernelGGL((sic_jump_kernel<false, true>), dim3(grid), dim3(512), 0, st, a, mdl);
    else hipLaunchKernelGGL((sic_jump_kernel<false, false>), dim3(grid), dim3(512), 0, st, a, mdl);
  }
}

void sic_launch_eval(const SicEvalArgs& a, const SicModel& mdl, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, sic_cus());
  if (mdl.cauchy) hipLaunchKernelGGL(sic_eval_kernel<true>, dim3(grid), dim3(512), 0, st, a, mdl);
  else hipLaunchKernelGGL(sic_eval_kernel<false>, dim3(grid), dim3(512), 0, st, a, mdl);
}

}  // namespace mjhmc
