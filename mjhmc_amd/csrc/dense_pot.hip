// ProductOfT (product of Student-t experts, mjhmc/misc/distributions.py:373-433) on the matrix cores.
//
//   u = (W^T x + b) / nu ,  E = sum_j (nu_j+1)/2 * log(1 + u_j^2) ,
//   dE/dx = W . ( (nu_j+1)/nu_j * u_j / (1 + u_j^2) )            (hand-derived; the reference uses T.grad)
//
// The reference evaluates this in float32 (float32 shared variables, allow_input_downcast=True, :398-415);
// so does this kernel: exact-f32 MFMA (v_mfma_f32_32x32x2_f32, bitwise an fmaf chain).
//
// One workgroup (4 waves, one per SIMD, up to 512 VGPRs each) owns a tile of 32 particles for the whole
// sampling_iteration.  Wave w owns rows [128w, 128w+128) of X, V, dEdX (and of U/H in expert space) for
// those 32 particles, held in REGISTERS in MFMA accumulator layout: lane = (particle c = lane&31,
// half h = lane>>5), register (block r, reg q) <-> row 128w + 4*((q&3) + 8*(q>>2) + 4h) + r.
// Two facts make the whole leapfrog run without re-shaping anything:
//   * an accumulator register IS a valid B operand of the next MFMA (k-pair = the two rows the two lane
//     halves hold) -- f32 needs no conversion -- so U -> H = phi(U) -> B of the second GEMM stays in place;
//   * GEMM outputs (rows = this wave's d-range, cols = particles) land exactly where X, V live, so the
//     kick/drift updates are register-local.
// Waves exchange only their B operands through LDS (publish 16 KB each, read the other three), two
// barriers per leapfrog step.  A operands (rows of the pre-scaled W / W^T copies, 1 MB each, L2-resident)
// are read straight from global memory, 512 contiguous bytes per lane half.
//
// Per leapfrog step and tile: 2 GEMMs x 1024 MFMAs per wave = 2 * 2 * 512 * 512 * 32 flop.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "dense_pot.hpp"

namespace mjhmc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kP = 32;    // particles per tile
constexpr int kDim = 512; // padded ndims == padded nbasis

// row held by (block r, reg q, lane half h) inside a wave's 128-row range
__device__ __forceinline__ int row_in_wave(int r, int q, int h) { return 4 * ((q & 3) + 8 * (q >> 2) + 4 * h) + r; }

struct Tile {
  f32x16 b[4];  // 4 blocks x 16 regs: 64 values per lane
};

// rows of particle `p` (particle-major [*, kDim] matrix): this lane's 16 groups of 4 consecutive dims
__device__ __forceinline__ void tile_load(const float* base, int64_t p, int w, int h, Tile& t) {
  const float* row = base + (size_t)p * kDim + 128 * w + 16 * h;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * ((q & 3) + 8 * (q >> 2)));
    t.b[0][q] = v[0];
    t.b[1][q] = v[1];
    t.b[2][q] = v[2];
    t.b[3][q] = v[3];
  }
}

__device__ __forceinline__ void tile_store(float* base, int64_t p, int w, int h, const Tile& t) {
  float* row = base + (size_t)p * kDim + 128 * w + 16 * h;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    f32x4 v;
    v[0] = t.b[0][q];
    v[1] = t.b[1][q];
    v[2] = t.b[2][q];
    v[3] = t.b[3][q];
    *reinterpret_cast<f32x4*>(row + 4 * ((q & 3) + 8 * (q >> 2))) = v;
  }
}

// LDS image of one wave's tile: [r][q/4][lane] x float4 (lane-linear 16 B: conflict-free b128)
using PubWave = f32x4[4][4][64];

__device__ __forceinline__ void publish(PubWave& dst, int lane, const Tile& t) {
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      f32x4 v;
      v[0] = t.b[r][4 * q4 + 0];
      v[1] = t.b[r][4 * q4 + 1];
      v[2] = t.b[r][4 * q4 + 2];
      v[3] = t.b[r][4 * q4 + 3];
      dst[r][q4][lane] = v;
    }
}

// acc[r][i][c] += sum over all 512 k-rows of  M[k][128w + 4i + r] * B[k][c]
//   M   : row-major [512][512] matrix in global memory (pre-scaled W or W^T)
//   pub : the four waves' published B tiles (k-rows in accumulator layout)
// The k-range is walked in 16 chunks of 16 k-pairs (one published block (ws, r) each).  With one wave
// per SIMD nothing else hides the L2 latency of the A rows, so they are software-pipelined by hand:
// chunk n+1's sixteen 16-byte A loads are issued (and fenced against sinking) before chunk n's 64 MFMAs
// (4096 matrix-pipe cycles) start.
__device__ __forceinline__ void a_chunk_load(const float* mlane, int chunk, f32x4 (&dst)[16]) {
  const float* base = mlane + (size_t)(128 * (chunk >> 2) + (chunk & 3)) * kDim;
#pragma unroll
  for (int q = 0; q < 16; ++q)
    dst[q] = *reinterpret_cast<const f32x4*>(base + (size_t)(4 * ((q & 3) + 8 * (q >> 2))) * kDim);
}

__device__ __forceinline__ void chunk_mfma(const PubWave* pub, int chunk, int lane, const f32x4 (&a)[16], Tile& acc) {
  const f32x4(*blk)[64] = pub[chunk >> 2][chunk & 3];
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const f32x4 b4 = blk[q4][lane];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const f32x4 a4 = a[4 * q4 + qq];
      acc.b[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0], b4[qq], acc.b[0], 0, 0, 0);
      acc.b[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[1], b4[qq], acc.b[1], 0, 0, 0);
      acc.b[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[2], b4[qq], acc.b[2], 0, 0, 0);
      acc.b[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[3], b4[qq], acc.b[3], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void gemm_512(const float* __restrict__ M, const PubWave* pub, int w, int c, int h, int lane,
                                         Tile& acc) {
  const float* mlane = M + (size_t)(16 * h) * kDim + 128 * w + 4 * c;
  f32x4 a0[16], a1[16];
  a_chunk_load(mlane, 0, a0);
#pragma unroll 1
  for (int chunk = 0; chunk < 16; chunk += 2) {
    a_chunk_load(mlane, chunk + 1, a1);
    __builtin_amdgcn_sched_barrier(0);
    chunk_mfma(pub, chunk, lane, a0, acc);
    __builtin_amdgcn_sched_barrier(0);
    a_chunk_load(mlane, chunk + 2 < 16 ? chunk + 2 : 15, a0);  // (the last one is a harmless re-read)
    __builtin_amdgcn_sched_barrier(0);
    chunk_mfma(pub, chunk + 1, lane, a1, acc);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// per-row constant vector in accumulator layout (row = 128w + row_in_wave(r, q, h))
__device__ __forceinline__ void rowvec_load(const float* vec, int w, int h, Tile& t) {
  const float* base = vec + 128 * w + 16 * h;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(base + 4 * ((q & 3) + 8 * (q >> 2)));
    t.b[0][q] = v[0];
    t.b[1][q] = v[1];
    t.b[2][q] = v[2];
    t.b[3][q] = v[3];
  }
}

// sum over this lane's 64 values and over the two lane halves -> per-particle partial of this wave
__device__ __forceinline__ float colsum(const Tile& t) {
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 16; ++q) s += t.b[r][q];
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(s), __float_as_int(s), false, false);
  return __int_as_float(sw[0]) + __int_as_float(sw[1]);
}

struct Shared {
  PubWave pub[2][4];     // 2 x 64 KB : B operands, [0] = X tile, [1] = H tile
  float red[2][4][kP];   // per-wave partial sums (energy, kinetic)
  float Hsum[2][kP];     // reduced
  int move[kP];          // transition chosen per particle
  float scal[4][kP];     // EXn, EVn, Hflf_out, spare
};

// gradient of the energy at the X held in `x`; optionally the energy itself.
// On return g holds dE/dX in the same layout as x.  Two barriers (X and H live in separate buffers:
// a wave can only reach the next publish of a buffer after every wave has passed the barrier that
// follows its last read of it).
__device__ __forceinline__ void pot_gradient(const PotModel& mdl, Shared& sh, int w, int c, int h, int lane,
                                             const Tile& x, Tile& g, bool want_energy, float* energy_out) {
  publish(sh.pub[0][w], lane, x);
  __syncthreads();
  Tile u;
  rowvec_load(mdl.cb, w, h, u);                   // u starts at b_j / nu_j
  gemm_512(mdl.W1, sh.pub[0], w, c, h, lane, u);  // + sum_d W[d][j]/nu_j * x_d
  if (want_energy) {                              // E = sum_j alpha_j log(1 + u_j^2)  (distributions.py:430-432)
    const float* al = mdl.alpha + 128 * w + 16 * h;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(al + 4 * ((q & 3) + 8 * (q >> 2)));
#pragma unroll
      for (int r = 0; r < 4; ++r) s += a4[r] * logf(1.0f + u.b[r][q] * u.b[r][q]);
    }
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(s), __float_as_int(s), false, false);
    const float part = __int_as_float(sw[0]) + __int_as_float(sw[1]);
    if (h == 0) sh.red[0][w][c] = part;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float uu = u.b[r][q];
      u.b[r][q] = uu / (1.0f + uu * uu);          // phi(u); the factor (nu+1)/nu lives in W2T
    }
  publish(sh.pub[1][w], lane, u);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 16; ++q) g.b[r][q] = 0.f;
  gemm_512(mdl.W2T, sh.pub[1], w, c, h, lane, g);
  if (want_energy && energy_out) {
    *energy_out = sh.red[0][0][c] + sh.red[0][1][c] + sh.red[0][2][c] + sh.red[0][3][c];
  }
}

// kinetic energy sum(v^2)/2 per particle (all lanes of column c get it).  One barrier pair.
__device__ __forceinline__ float pot_kinetic(Shared& sh, int w, int c, int h, const Tile& v) {
  Tile s;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int q = 0; q < 16; ++q) s.b[r][q] = v.b[r][q] * v.b[r][q];
  const float part = colsum(s);
  if (h == 0) sh.red[1][w][c] = part;
  __syncthreads();
  const float tot = (sh.red[1][0][c] + sh.red[1][1][c] + sh.red[1][2][c] + sh.red[1][3][c]) / 2.0f;
  __syncthreads();
  return tot;
}

// L leapfrog steps (hmc_state.py:86-100); g enters as dE/dX at x, leaves as dE/dX at the new x.
// Returns E(x_new) through *ex (the last gradient evaluation already has u(x_new)).
__device__ __forceinline__ void pot_trajectory(const PotModel& mdl, Shared& sh, int w, int c, int h, int lane, Tile& x,
                                               Tile& v, Tile& g, int L, float eps, float chalf, float* ex) {
  for (int s = 0; s < L; ++s) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        v.b[r][q] = v.b[r][q] + chalf * g.b[r][q];
        x.b[r][q] = x.b[r][q] + eps * v.b[r][q];
      }
    pot_gradient(mdl, sh, w, c, h, lane, x, g, s == L - 1, ex);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) v.b[r][q] = v.b[r][q] + chalf * g.b[r][q];
  }
}

// ---------------------------------------------------------------------------------------------------
// evaluation: E(X), dEdX(X), optional kinetic energy / generated momentum (HMCState.__init__)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void pot_eval_kernel(const PotEvalArgs a, const PotModel mdl) {
  __shared__ Shared sh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t p = tile * kP + c;
    Tile x, g;
    tile_load(a.X, p, w, h, x);
    float ex = 0.f;
    pot_gradient(mdl, sh, w, c, h, lane, x, g, true, &ex);
    if (a.G) tile_store(a.G, p, w, h, g);
    if (a.E && w == 0 && h == 0) a.E[p] = ex;
    if (a.EV) {
      Tile v;
      if (a.V_gen) {
        // tick-0 momentum: Box-Muller pairs (dims 2k, 2k+1) of the counter RNG
#pragma unroll 1
        for (int q = 0; q < 16; ++q) {
          const int d = 128 * w + row_in_wave(0, q, h);
          double z0, z1, z2, z3;
          const uint32_t pid = (uint32_t)(a.first_pid + (p < a.N ? p : 0));
          normal_pair(a.key, pid, (uint32_t)(d >> 1), z0, z1);
          normal_pair(a.key, pid, (uint32_t)((d >> 1) + 1), z2, z3);
          v.b[0][q] = d + 0 < a.D ? (float)z0 : 0.f;
          v.b[1][q] = d + 1 < a.D ? (float)z1 : 0.f;
          v.b[2][q] = d + 2 < a.D ? (float)z2 : 0.f;
          v.b[3][q] = d + 3 < a.D ? (float)z3 : 0.f;
        }
        tile_store(a.V_gen, p, w, h, v);
      } else {
        tile_load(a.V, p, w, h, v);
      }
      const float ev = pot_kinetic(sh, w, c, h, v);
      if (w == 0 && h == 0) a.EV[p] = ev;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// cold-cache compaction.  A tile of 32 particles costs the same whether 1 or 32 of them need the
// inverse-L trajectory, so the cold particles (5-60 % of the batch) are gathered into dense tiles
// first: the F L F work stays proportional to the cold fraction, as in the reference (hmc_state.py:109-119).
// ---------------------------------------------------------------------------------------------------
__global__ void pot_cold_list_kernel(const float* __restrict__ Hflf_in, float* __restrict__ Hwork, int64_t N,
                                     int64_t Npad, int* __restrict__ list, int* __restrict__ count,
                                     const Control* ctl, unsigned long long* stats) {
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

__global__ __launch_bounds__(256, 1) void pot_flf_kernel(const PotJumpArgs a, const PotModel mdl) {
  __shared__ Shared sh;
  if (a.ctl->failed) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  const int ncold = *a.cold_count;
  for (int tile = blockIdx.x; tile * kP < ncold; tile += gridDim.x) {
    const int slot = tile * kP + c;
    const int64_t p = a.cold_list[slot < ncold ? slot : ncold - 1];  // pad the last tile with a repeat
    Tile x, v, g;
    tile_load(a.X_in, p, w, h, x);
    tile_load(a.V_in, p, w, h, v);
    tile_load(a.G_in, p, w, h, g);
#pragma unroll
    for (int r = 0; r < 4; ++r) v.b[r] = -v.b[r];
    float ex = 0.f;
    pot_trajectory(mdl, sh, w, c, h, lane, x, v, g, a.L, a.eps, a.chalf, &ex);
    const float ev = pot_kinetic(sh, w, c, h, v);
    if (w == 0 && h == 0) a.Hwork[p] = ex + ev;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// the jump kernel (MJHMC mode): one sampling_iteration attempt for a tile of 32 particles
// ---------------------------------------------------------------------------------------------------
template <bool REPLAY>
__global__ __launch_bounds__(256, 1) void pot_jump_kernel(const PotJumpArgs a, const PotModel mdl) {
  __shared__ Shared sh;
  if (a.ctl->failed) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  unsigned nL = 0, nF = 0, nR = 0, nCold = 0;
  bool any_bad = false;
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t p = tile * kP + c;
    const bool alive = p < a.N;
    const float EX0 = a.EX_in[p], EV0 = a.EV_in[p], Hc = a.Hflf_in[p];
    const float H0 = EX0 + EV0;
    const bool warm = (Hc == Hc) || !alive;
    // H of the inverse-L proposal: cached, or integrated by pot_flf_kernel for the cold particles
    const float Hflf = a.Hwork[p];
    Tile x, v, g;
    tile_load(a.X_in, p, w, h, x);
    tile_load(a.V_in, p, w, h, v);
    tile_load(a.G_in, p, w, h, g);
    float EXL = 0.f;
    pot_trajectory(mdl, sh, w, c, h, lane, x, v, g, a.L, a.eps, a.chalf, &EXL);
    const float EVL = pot_kinetic(sh, w, c, h, v);
    const float HL = EXL + EVL;

    // rates, waiting times, first minimum: lanes 0..31 of wave 0, one particle each
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
      // scalars of the L and F successors; R's kinetic energy is filled in below
      a.EX_out[p] = (k == 0) ? EXL : EX0;
      a.EV_out[p] = (k == 0) ? EVL : EV0;
      a.Hflf_out[p] = (k == 0) ? H0 : __builtin_nanf("");
    }
    __syncthreads();
    const int k = sh.move[c];
    const bool tile_has_r = __ballot(k == 2) != 0ull;
    if (k != 0) {  // F / R keep the position (and its gradient)
      tile_load(a.X_in, p, w, h, x);
      tile_load(a.G_in, p, w, h, g);
      Tile v0;
      tile_load(a.V_in, p, w, h, v0);
#pragma unroll
      for (int r = 0; r < 4; ++r) v.b[r] = -v0.b[r];
      if (k == 2) {  // HMCState.R (hmc_state.py:121-129)
        const uint32_t pid = (uint32_t)(a.first_pid + (alive ? p : 0));
        Tile z;
        if constexpr (REPLAY) {
          tile_load(a.noise, alive ? p : 0, w, h, z);
        } else {
#pragma unroll 1
          for (int q = 0; q < 16; ++q) {
            const int d = 128 * w + row_in_wave(0, q, h);
            double z0, z1, z2, z3;
            normal_pair(a.key, pid, (uint32_t)(d >> 1), z0, z1);
            normal_pair(a.key, pid, (uint32_t)((d >> 1) + 1), z2, z3);
            z.b[0][q] = d + 0 < a.D ? (float)z0 : 0.f;
            z.b[1][q] = d + 1 < a.D ? (float)z1 : 0.f;
            z.b[2][q] = d + 2 < a.D ? (float)z2 : 0.f;
            z.b[3][q] = d + 3 < a.D ? (float)z3 : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v.b[r] = v0.b[r] * a.r_keep + z.b[r] * a.r_mix;
      }
    }
    if (tile_has_r) {  // all waves take part in the reduction; only R columns use the result
      const float evr = pot_kinetic(sh, w, c, h, v);
      if (k == 2 && w == 0 && h == 0) a.EV_out[p] = evr;
    }
    tile_store(a.X_out, p, w, h, x);
    tile_store(a.V_out, p, w, h, v);
    tile_store(a.G_out, p, w, h, g);
    __syncthreads();
  }
  if (any_bad) {
    a.ctl->failed = 1;
    a.ctl->failed_iter = a.iter;
  }
  __shared__ unsigned tally[4];
  if (threadIdx.x < 4) tally[threadIdx.x] = 0;
  __syncthreads();
  if (nL) atomicAdd(&tally[0], nL);
  if (nF) atomicAdd(&tally[1], nF);
  if (nR) atomicAdd(&tally[2], nR);
  if (nCold) atomicAdd(&tally[3], nCold);
  __syncthreads();
  if (threadIdx.x < 4 && tally[threadIdx.x]) atomicAdd(&a.stats[threadIdx.x], (unsigned long long)tally[threadIdx.x]);
}

static int resident_cus() {
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return std::max(1, cus);
}

void pot_launch_jump(const PotJumpArgs& a, const PotModel& mdl, hipStream_t st) {
  (void)hipMemsetAsync(a.cold_count, 0, sizeof(int), st);
  hipLaunchKernelGGL(pot_cold_list_kernel, dim3((unsigned)((a.Npad + 255) / 256)), dim3(256), 0, st, a.Hflf_in, a.Hwork,
                     a.N, a.Npad, a.cold_list, a.cold_count, (const Control*)a.ctl, a.stats);
  hipLaunchKernelGGL(pot_flf_kernel, dim3((unsigned)std::min<int64_t>(a.ntiles, resident_cus())), dim3(256), 0, st, a, mdl);
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, resident_cus());
  if (a.rexp && a.noise) hipLaunchKernelGGL(pot_jump_kernel<true>, dim3(grid), dim3(256), 0, st, a, mdl);
  else hipLaunchKernelGGL(pot_jump_kernel<false>, dim3(grid), dim3(256), 0, st, a, mdl);
}

void pot_launch_eval(const PotEvalArgs& a, const PotModel& mdl, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, resident_cus());
  hipLaunchKernelGGL(pot_eval_kernel, dim3(grid), dim3(256), 0, st, a, mdl);
}

}  // namespace mjhmc
