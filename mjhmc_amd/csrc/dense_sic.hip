// SparseImageCode (sparse-coding posterior over coefficients, mjhmc/misc/tf_distributions.py:204-272) on
// the bf16 matrix cores: bf16 state in HBM, bf16 MFMA operands, fp32 accumulation and fp32 integrator
// registers (BASELINE.json configs[4]).  n_coeffs = 1024, img_size = 256, P = n_patches patches per particle (the
// reference's default is 9, tf_distributions.py:208; BASELINE configs[4] is one):
//
//   resid_p = B a_p - y_p ,  E = mean_p 1/2 |resid_p|^2 + lambda * sum log(1 + a^2)   (Cauchy prior, :259-267)
//   dE/da_p = B^T resid_p / P + lambda * 2a / (1 + a^2)                               (or lambda * sign(a), Laplace)
//
// A particle's state row is its P coefficient vectors back to back (patch-major, :249 for one active column), i.e.
// P consecutive 1024-rows of the (N * P, 1024) matrix: every (particle, patch) pair is a COLUMN of the two GEMMs, and
// only the energy sums and the jump decision couple the P columns of a particle.  A tile is 32 columns holding
// floor(32 / P) whole particles (P = 9: 27 of the 32 columns work).
//
// One workgroup = 8 waves (two per SIMD) owns a tile of 32 columns.  Wave w holds coefficient rows
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
  float colsum[kP];         // per-column energies, summed over a particle's columns when n_patches > 1
  int move[kP];
};

// what column c of a tile works on
struct Col {
  int64_t part;  // particle (clamped to a valid one when the column idles)
  int64_t q;     // row of the (N * P, 1024) coefficient matrix: part * P + patch
  int patch;
  int g0;        // first column of this particle's group
  bool alive;    // the column holds a real (particle, patch) pair: its results are stored
  bool leader;   // first column of a live particle: writes the per-particle scalars and tallies
};

// slot = index of the column's particle among the tile's particles; `part_of(slot)` resolves it (identity for the jump
// kernel, a lookup in the cold list for the inverse-L pass)
template <class PartOf>
__device__ __forceinline__ Col col_of(int64_t tile, int c, int P, int64_t n_parts, const PartOf& part_of) {
  const int ppt = kP / P, cpt = ppt * P;
  const int cc = c < cpt ? c : cpt - 1;
  const int64_t slot = tile * ppt + cc / P;
  Col k;
  k.alive = (c < cpt) && slot < n_parts;
  k.part = part_of(slot < n_parts ? slot : n_parts - 1);
  k.patch = cc % P;
  k.q = k.part * P + k.patch;
  k.g0 = (cc / P) * P;
  k.leader = k.alive && k.patch == 0;
  return k;
}
struct Identity {
  __device__ int64_t operator()(int64_t s) const { return s; }
};

__device__ __forceinline__ f32x4 frag_of(const f32x16& acc, int s, float scale) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (__bf16)(acc[8 * s + j] * scale);
  return __builtin_bit_cast(f32x4, f);
}

// residual of the tile at the X held in x (GEMM1).  Leaves it in `res` (fp32 accumulator layout).
__device__ __forceinline__ void sic_residual(const SicModel& mdl, SicShared& sh, int w, int c, int h, int lane, int patch,
                                             const CTile& x, f32x16& res) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    sh.pubA[w][b][0][lane] = frag_of(x.b[b], 0, 1.0f);
    sh.pubA[w][b][1][lane] = frag_of(x.b[b], 1, 1.0f);
  }
  __syncthreads();
  {  // res starts at -y[i]
    const float* yv = mdl.y + kI * patch + 32 * w + 4 * h;
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
  sh.pubR[w][0][lane] = frag_of(res, 0, scale * mdl.invP);  // d/da_p of the MEAN over patches
  sh.pubR[w][1][lane] = frag_of(res, 1, scale * mdl.invP);
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

// sum of a per-column value over the columns of the caller's particle (n_patches of them, consecutive)
__device__ __forceinline__ float group_total(SicShared& sh, int w, int c, int h, int P, int g0, float col_tot) {
  if (P == 1) return col_tot;
  if (w == 0 && h == 0) sh.colsum[c] = col_tot;
  __syncthreads();
  float t = 0.f;
  for (int j = 0; j < P; ++j) t += sh.colsum[g0 + j];
  __syncthreads();
  return t;
}

// E(x) per PARTICLE from the residuals at x:  mean_p 1/2 |res_p|^2 + lambda * prior(x)  (tf_distributions.py:257-270)
template <bool CAUCHY>
__device__ __forceinline__ float sic_energy(const SicModel& mdl, SicShared& sh, int w, int c, int h, const Col& col,
                                            const f32x16& res, const CTile& x) {
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
  const float part = half_swap_sum(s * mdl.invP + mdl.lambda * pr);
  if (h == 0) sh.red[0][w][c] = part;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) tot += sh.red[0][k][c];
  __syncthreads();
  return group_total(sh, w, c, h, mdl.P, col.g0, tot);
}

// sum(v^2) / 2 per PARTICLE
__device__ __forceinline__ float sic_kinetic(SicShared& sh, int w, int c, int h, int P, const Col& col, const CTile& v) {
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
  return group_total(sh, w, c, h, P, col.g0, tot) / 2.0f;
}

// L leapfrog steps with the half kicks between drifts merged (bf16 operands make the reference's
// separate roundings meaningless).  Returns E(x_new) of the particle; x, v updated in place.
template <bool CAUCHY>
__device__ __forceinline__ float sic_trajectory(const SicModel& mdl, SicShared& sh, int w, int c, int h, int lane,
                                                const Col& col, CTile& x, CTile& v, int L, float eps, float chalf) {
  f32x16 res;
  sic_residual(mdl, sh, w, c, h, lane, col.patch, x, res);
  if (L > 0) sic_kick<CAUCHY>(mdl, sh, w, c, h, lane, res, x, chalf, v);
  for (int s = 1; s <= L; ++s) {
#pragma unroll
    for (int b = 0; b < 4; ++b) x.b[b] = x.b[b] + eps * v.b[b];
    sic_residual(mdl, sh, w, c, h, lane, col.patch, x, res);
    sic_kick<CAUCHY>(mdl, sh, w, c, h, lane, res, x, s < L ? 2.0f * chalf : chalf, v);
  }
  // the successor position is stored in bf16 (and GEMM1 already saw bf16(x)): evaluate the prior on
  // what will be stored, so EX is the energy of the stored state
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int q = 0; q < 16; ++q) x.b[b][q] = (float)(__bf16)x.b[b][q];
  return sic_energy<CAUCHY>(mdl, sh, w, c, h, col, res, x);
}

__device__ __forceinline__ void round_to_state(CTile& t) {
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int q = 0; q < 16; ++q) t.b[b][q] = (float)(__bf16)t.b[b][q];
}

// v += mix * (standard normals of this lane's dims of particle `pid`; dims patch * 1024 + ...), group by group: one
// Box-Muller quadruple's temporaries at a time keeps this rare branch from dictating the kernel's register budget
__device__ __forceinline__ void sic_add_normals(const RngKey& key, uint32_t pid, int patch, int w, int h, float mix,
                                                CTile& v) {
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll 1
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d = kC * patch + 128 * w + 32 * b + 8 * g4 + 4 * h;
      float z0, z1, z2, z3;
      normal_pair_f32(key, pid, (uint32_t)(d >> 1), z0, z1);
      normal_pair_f32(key, pid, (uint32_t)((d >> 1) + 1), z2, z3);
      f32x4 add;
      add[0] = z0 * mix;
      add[1] = z1 * mix;
      add[2] = z2 * mix;
      add[3] = z3 * mix;
      // g4 is a runtime index here: address the four registers through selects
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if ((q >> 2) == g4) v.b[b][q] += add[q & 3];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
template <bool CAUCHY>
__global__ __launch_bounds__(512, 2) void sic_eval_kernel(const SicEvalArgs a, const SicModel mdl) {
  __shared__ SicShared sh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const Col col = col_of(tile, c, mdl.P, a.N, Identity{});
    CTile x;
    ctile_load(a.X, col.q, w, h, x);
    f32x16 res;
    sic_residual(mdl, sh, w, c, h, lane, col.patch, x, res);
    if (a.G) {
      CTile g;
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) g.b[b][q] = 0.f;
      sic_kick<CAUCHY>(mdl, sh, w, c, h, lane, res, x, 1.0f, g);
      if (col.alive) {
        float* row = a.G + (size_t)col.q * kC + 128 * w + 4 * h;  // dE/dX is handed out in float32
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
    }
    const float ex = sic_energy<CAUCHY>(mdl, sh, w, c, h, col, res, x);
    if (a.E && w == 0 && h == 0 && col.leader) a.E[col.part] = ex;
    if (a.EV) {
      CTile v;
      if (a.V_gen) {
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int q = 0; q < 16; ++q) v.b[b][q] = 0.f;
        sic_add_normals(a.key, (uint32_t)(a.first_pid + col.part), col.patch, w, h, 1.0f, v);
        round_to_state(v);  // EV matches the stored momentum
        if (col.alive) ctile_store(a.V_gen, col.q, w, h, v);
      } else {
        ctile_load(a.V, col.q, w, h, v);
      }
      const float ev = sic_kinetic(sh, w, c, h, mdl.P, col, v);
      if (w == 0 && h == 0 && col.leader) a.EV[col.part] = ev;
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

struct FromList {
  const int* list;
  __device__ int64_t operator()(int64_t s) const { return list[s]; }
};

template <bool CAUCHY>
__global__ __launch_bounds__(512, 2) void sic_flf_kernel(const SicJumpArgs a, const SicModel mdl) {
  __shared__ SicShared sh;
  if (a.ctl->failed) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  const int ncold = *a.cold_count;
  const int ppt = kP / mdl.P;
  for (int64_t tile = blockIdx.x; tile * ppt < ncold; tile += gridDim.x) {
    const Col col = col_of(tile, c, mdl.P, (int64_t)ncold, FromList{a.cold_list});  // the last tile repeats an entry
    CTile x, v;
    ctile_load(a.X_in, col.q, w, h, x);
    ctile_load(a.V_in, col.q, w, h, v);
#pragma unroll
    for (int b = 0; b < 4; ++b) v.b[b] = -v.b[b];
    const float ex = sic_trajectory<CAUCHY>(mdl, sh, w, c, h, lane, col, x, v, a.L, a.eps, a.chalf);
    round_to_state(v);  // the same rounding the jump kernel applies to the forward proposal
    const float ev = sic_kinetic(sh, w, c, h, mdl.P, col, v);
    if (w == 0 && h == 0 && col.leader) a.Hwork[col.part] = ex + ev;
    __syncthreads();
  }
}

// MODE = kModeMJHMC (markov_jump_hmc.py:355-415), kModeCT (:251-290) or kModeControl (:116-148, the comparison arm of
// the reference's sparse-coding experiments, search/control_sp_img/control_objective.py:10)
template <bool CAUCHY, bool REPLAY, int MODE>
__global__ __launch_bounds__(512, 2) void sic_jump_kernel(const SicJumpArgs a, const SicModel mdl) {
  __shared__ SicShared sh;
  if (a.ctl->failed) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0;  // tallies (meaning per mode: fill_iter_stats in api.hip)
  bool any_bad = false;
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const Col col = col_of(tile, c, mdl.P, a.N, Identity{});
    const int64_t p = col.part;
    const float EX0 = a.EX_in[p], EV0 = a.EV_in[p];
    const float H0 = EX0 + EV0;
    const float Hflf = MODE == kModeMJHMC ? a.Hwork[p] : 0.f;
    CTile x, v;
    ctile_load(a.X_in, col.q, w, h, x);
    ctile_load(a.V_in, col.q, w, h, v);
    const float EXL = sic_trajectory<CAUCHY>(mdl, sh, w, c, h, lane, col, x, v, a.L, a.eps, a.chalf);
    round_to_state(v);  // the successor state is stored in bf16: report the kinetic energy of what is stored
    const float EVL = sic_kinetic(sh, w, c, h, mdl.P, col, v);
    const float HL = EXL + EVL;
    if (w == 0 && h == 0) {  // every column of a particle reaches the same decision; its leader reports it
      const uint32_t pid = (uint32_t)(a.first_pid + p);
      double best = 0.0;
      bool bad = false, gate = false;
      int k;
      if constexpr (MODE == kModeMJHMC) k = dense_decide<REPLAY>(H0, HL, Hflf, a.p_r, pid, p, a.N, a.rexp, a.key, best, bad);
      else if constexpr (MODE == kModeCT) k = dense_decide_ct<REPLAY>(H0, HL, a.p_r, pid, p, a.N, a.rexp, a.key, best, bad);
      else k = dense_control<REPLAY>(H0, HL, a.p_r, a.p_flip, pid, p, a.N, a.runif, a.key, gate);
      sh.move[c] = k | (gate ? 4 : 0);
      if (col.leader) {
        any_bad |= bad;
        a.dwell[p] = best;
        a.dwell_ring[p] = best;
        a.trans[p] = (uint8_t)k;
        if constexpr (MODE == kModeControl) {  // l_count, f_count, R applied, fl_count (markov_jump_hmc.py:143-148)
          n0 += (k == 3);
          n1 += (k == 2);
          n2 += gate ? 1u : 0u;
          n3 += (k == 1);
        } else {
          n0 += (k == 0);
          n1 += (k == 1);
          n2 += (k == 2);
        }
        const bool took_L = MODE == kModeControl ? (k & 1) : (k == 0);
        a.EX_out[p] = took_L ? EXL : EX0;
        a.EV_out[p] = took_L ? EVL : EV0;
        a.Hflf_out[p] = (MODE == kModeMJHMC && k == 0) ? H0 : __builtin_nanf("");
      }
    }
    __syncthreads();
    const int mv = sh.move[c];
    const int k = mv & 3;
    bool refresh;  // this column's momentum is redrawn (HMCState.R)
    if constexpr (MODE == kModeControl) {
      if (!(k & 1)) {  // rejected: back to the pre-move state
        ctile_load(a.X_in, col.q, w, h, x);
        ctile_load(a.V_in, col.q, w, h, v);
      } else {  // accepted L F: flip
#pragma unroll
        for (int b = 0; b < 4; ++b) v.b[b] = -v.b[b];
      }
      if (k & 2) {
#pragma unroll
        for (int b = 0; b < 4; ++b) v.b[b] = -v.b[b];
      }
      refresh = (mv & 4) != 0;  // batch-wide (markov_jump_hmc.py:138-141)
    } else {
      if (k != 0) {  // F / R keep the position
        ctile_load(a.X_in, col.q, w, h, x);
        ctile_load(a.V_in, col.q, w, h, v);
      }
      if ((MODE == kModeCT && k == 0) || k == 1) {  // CT's FL move ends with a flip (:258,278); F flips
#pragma unroll
        for (int b = 0; b < 4; ++b) v.b[b] = -v.b[b];
      }
      refresh = (k == 2);
    }
    const bool tile_refreshes = __ballot(refresh) != 0ull;
    if (refresh) {  // HMCState.R (hmc_state.py:121-129)
#pragma unroll
      for (int b = 0; b < 4; ++b) v.b[b] = v.b[b] * a.r_keep;
      if constexpr (REPLAY) {
        const __bf16* zrow = a.noise + (size_t)col.q * kC + 128 * w + 4 * h;
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const bf16x4 z = *reinterpret_cast<const bf16x4*>(zrow + 32 * b + 8 * g4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) v.b[b][4 * g4 + kk] += (float)z[kk] * a.r_mix;
          }
      } else {
        sic_add_normals(a.key, (uint32_t)(a.first_pid + p), col.patch, w, h, a.r_mix, v);
      }
      round_to_state(v);
    }
    if (tile_refreshes) {
      const float evr = sic_kinetic(sh, w, c, h, mdl.P, col, v);
      if (refresh && w == 0 && h == 0 && col.leader) a.EV_out[p] = evr;
    }
    if (col.alive) {
      ctile_store(a.X_out, col.q, w, h, x);
      ctile_store(a.V_out, col.q, w, h, v);
    }
    __syncthreads();
  }
  if (any_bad) {
    a.ctl->failed = 1;
    a.ctl->failed_iter = a.iter;
  }
  __shared__ unsigned tally[4];
  if (threadIdx.x < 4) tally[threadIdx.x] = 0;
  __syncthreads();
  if (n0) atomicAdd(&tally[0], n0);
  if (n1) atomicAdd(&tally[1], n1);
  if (n2) atomicAdd(&tally[2], n2);
  if (n3) atomicAdd(&tally[3], n3);
  __syncthreads();
  if (threadIdx.x < 4 && tally[threadIdx.x]) atomicAdd(&a.stats[threadIdx.x], (unsigned long long)tally[threadIdx.x]);
}

// HMCState.leapfrog / HMCState.L on caller-supplied states (hmc_state.py:86-100)
template <bool CAUCHY>
__global__ __launch_bounds__(512, 2) void sic_leap_kernel(const SicLeapArgs a, const SicModel mdl) {
  __shared__ SicShared sh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const Col col = col_of(tile, c, mdl.P, a.N, Identity{});
    CTile x, v;
    ctile_load(a.X, col.q, w, h, x);
    ctile_load(a.V, col.q, w, h, v);
    const float ex = sic_trajectory<CAUCHY>(mdl, sh, w, c, h, lane, col, x, v, a.L, a.eps, a.chalf);
    round_to_state(v);
    const float ev = sic_kinetic(sh, w, c, h, mdl.P, col, v);
    if (a.G) {  // dE/dX of the stored end point
      f32x16 res;
      sic_residual(mdl, sh, w, c, h, lane, col.patch, x, res);
      CTile g;
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) g.b[b][q] = 0.f;
      sic_kick<CAUCHY>(mdl, sh, w, c, h, lane, res, x, 1.0f, g);
      if (col.alive) {
        float* row = a.G + (size_t)col.q * kC + 128 * w + 4 * h;
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
    }
    if (col.alive) {
      ctile_store(a.X_out, col.q, w, h, x);
      ctile_store(a.V_out, col.q, w, h, v);
    }
    if (w == 0 && h == 0 && col.leader) {
      if (a.EX) a.EX[col.part] = ex;
      if (a.EV) a.EV[col.part] = ev;
    }
    __syncthreads();
  }
}

static int sic_cus() {
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return std::max(1, cus);
}

template <bool CAUCHY, int MODE>
static void sic_launch_mode(const SicJumpArgs& a, const SicModel& mdl, unsigned grid, hipStream_t st) {
  const bool replay = MODE == kModeControl ? (a.runif && a.noise) : (a.rexp && a.noise);
  if (replay) hipLaunchKernelGGL((sic_jump_kernel<CAUCHY, true, MODE>), dim3(grid), dim3(512), 0, st, a, mdl);
  else hipLaunchKernelGGL((sic_jump_kernel<CAUCHY, false, MODE>), dim3(grid), dim3(512), 0, st, a, mdl);
}

template <bool CAUCHY>
static void sic_launch_jump_t(const SicJumpArgs& a, const SicModel& mdl, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, sic_cus());
  if (a.mode == kModeMJHMC) {  // only MJHMC has the inverse-L proposal and its cache
    (void)hipMemsetAsync(a.cold_count, 0, sizeof(int), st);
    hipLaunchKernelGGL(sic_cold_list_kernel, dim3((unsigned)((a.Npad + 255) / 256)), dim3(256), 0, st, a.Hflf_in, a.Hwork,
                       a.N, a.Npad, a.cold_list, a.cold_count, (const Control*)a.ctl, a.stats);
    hipLaunchKernelGGL(sic_flf_kernel<CAUCHY>, dim3(grid), dim3(512), 0, st, a, mdl);
    sic_launch_mode<CAUCHY, kModeMJHMC>(a, mdl, grid, st);
  } else if (a.mode == kModeCT) {
    sic_launch_mode<CAUCHY, kModeCT>(a, mdl, grid, st);
  } else {
    sic_launch_mode<CAUCHY, kModeControl>(a, mdl, grid, st);
  }
}

void sic_launch_jump(const SicJumpArgs& a, const SicModel& mdl, hipStream_t st) {
  if (mdl.cauchy) sic_launch_jump_t<true>(a, mdl, st);
  else sic_launch_jump_t<false>(a, mdl, st);
}

void sic_launch_eval(const SicEvalArgs& a, const SicModel& mdl, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, sic_cus());
  if (mdl.cauchy) hipLaunchKernelGGL(sic_eval_kernel<true>, dim3(grid), dim3(512), 0, st, a, mdl);
  else hipLaunchKernelGGL(sic_eval_kernel<false>, dim3(grid), dim3(512), 0, st, a, mdl);
}

void sic_launch_leap(const SicLeapArgs& a, const SicModel& mdl, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, sic_cus());
  if (mdl.cauchy) hipLaunchKernelGGL(sic_leap_kernel<true>, dim3(grid), dim3(512), 0, st, a, mdl);
  else hipLaunchKernelGGL(sic_leap_kernel<false>, dim3(grid), dim3(512), 0, st, a, mdl);
}

}  // namespace mjhmc
