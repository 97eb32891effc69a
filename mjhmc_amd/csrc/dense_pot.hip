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
// (Sizes below are for the 512-dim build, NB = 4; ndims <= 128 / 256 use NB = 1 / 2 with the same code.)
// Waves exchange only their B operands through LDS (publish 16 KB each, read the other three), two
// barriers per leapfrog step.  A operands (rows of the pre-scaled W / W^T copies, 1 MB each, L2-resident)
// are read straight from global memory, 512 contiguous bytes per lane half.
//
// Per leapfrog step and tile: 2 GEMMs x 1024 MFMAs per wave = 2 * 2 * 512 * 512 * 32 flop.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "dense_pot.hpp"
#include "dense_pot_tile.hpp"

namespace mjhmc {

// ---------------------------------------------------------------------------------------------------
// evaluation: E(X), dEdX(X), optional kinetic energy / generated momentum (HMCState.__init__)
// ---------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256, 1) void pot_eval_kernel(const PotEvalArgs a, const PotModel mdl) {
  __shared__ Shared<NB> sh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  AReg<NB> ar;
  areg_load<NB>(mdl, w, c, h, ar);
  stage_bias<NB>(mdl, sh);
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t p = tile * kP + c;
    Tile<NB> x, g;
    tile_load<NB>(a.X, p, w, h, x);
    float ex = 0.f;
    pot_gradient<NB>(mdl, ar, sh, w, c, h, lane, x, g, true, &ex);
    if (a.G) tile_store<NB>(a.G, p, w, h, g);
    if (a.E && w == 0 && h == 0) a.E[p] = ex;
    if (a.EV) {
      Tile<NB> v;
      if (a.V_gen) {
        pot_normals<NB>(a.key, (uint32_t)(a.first_pid + (p < a.N ? p : 0)), w, h, a.D, v);
        tile_store<NB>(a.V_gen, p, w, h, v);
      } else {
        tile_load<NB>(a.V, p, w, h, v);
      }
      const float ev = pot_kinetic<NB>(sh, w, c, h, v);
      if (w == 0 && h == 0) a.EV[p] = ev;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// cold-cache compaction.  A tile of 32 particles costs the same whether 1 or 32 of them need the
// inverse-L trajectory, so the cold particles (5-60 % of the batch) are gathered into dense tiles
// first: the F L F work stays proportional to the cold fraction, as in the reference (hmc_state.py:109-119).
// ---------------------------------------------------------------------------------------------------
// The list of the FIRST iteration of a mjhmc_iterate call comes from a scan of the cache (this kernel, once per call);
// every later one is written by the jump kernel of the iteration before it (append_cold): no per-iteration memset, no
// per-iteration list kernel -- a 4-byte memset behind a persistent grid waited for a workgroup to exit (20 % of the
// summed kernel time of a C3 profile was such queue waits).
__global__ void pot_cold_list_kernel(const float* __restrict__ Hflf_in, int64_t N, int* __restrict__ list,
                                     int* __restrict__ count, const Control* ctl) {
  if (ctl->failed) return;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float hc = p < N ? Hflf_in[p] : 0.f;
  append_cold(list, count, (p < N) && !(hc == hc), p);
}

template <int NB>
__global__ __launch_bounds__(256, 1) void pot_flf_kernel(const PotJumpArgs a, const PotModel mdl) {
  __shared__ Shared<NB> sh;
  if (a.ctl->failed) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  const int ncold = *a.cold_count;
  if (blockIdx.x == 0 && threadIdx.x == 0 && ncold) atomicAdd(&a.stats[3], (unsigned long long)ncold);  // the cold tally
  if ((int64_t)blockIdx.x * kP >= ncold) return;  // nothing for this workgroup
  AReg<NB> ar;
  areg_load<NB>(mdl, w, c, h, ar);
  stage_bias<NB>(mdl, sh);
  for (int tile = blockIdx.x; tile * kP < ncold; tile += gridDim.x) {
    const int slot = tile * kP + c;
    const int64_t p = a.cold_list[slot < ncold ? slot : ncold - 1];  // pad the last tile with a repeat
    Tile<NB> x, v, g;
    tile_load<NB>(a.X_in, p, w, h, x);
    tile_load<NB>(a.V_in, p, w, h, v);
    tile_load<NB>(a.G_in, p, w, h, g);
#pragma unroll
    for (int r = 0; r < NB; ++r) v.b[r] = -v.b[r];
    float ex = 0.f;
    pot_trajectory<NB>(mdl, ar, sh, w, c, h, lane, x, v, g, a.L, a.eps, a.chalf, &ex);
    const float ev = pot_kinetic<NB>(sh, w, c, h, v);
    if (w == 0 && h == 0) a.Hwork[p] = ex + ev;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// the jump kernel: one sampling_iteration attempt for a tile of 32 particles.
// MODE = kModeMJHMC (markov_jump_hmc.py:355-415), kModeCT (ContinuousTimeHMC, :251-290) or kModeControl (HMCBase /
// HMC / ControlHMC, :116-148 -- the comparison arm of the reference's ProductOfT experiments,
// search/control_poe_36/mjhmc_objective.py:14).
// ---------------------------------------------------------------------------------------------------
template <int NB, bool REPLAY, int MODE>
__global__ __launch_bounds__(256, 1) void pot_jump_kernel(const PotJumpArgs a, const PotModel mdl) {
  __shared__ Shared<NB> sh;
  if (a.ctl->failed) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  // the inverse-L pass of this iteration has consumed its list: its counter is free for the iteration after the next
  if (MODE == kModeMJHMC && blockIdx.x == 0 && threadIdx.x == 0) *a.cold_count = 0;
  unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0;  // tallies (meaning per mode: fill_iter_stats in api.hip)
  bool any_bad = false;
  AReg<NB> ar;
  areg_load<NB>(mdl, w, c, h, ar);
  stage_bias<NB>(mdl, sh);
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t p = tile * kP + c;
    const bool alive = p < a.N;
    const float EX0 = a.EX_in[p], EV0 = a.EV_in[p];
    const float H0 = EX0 + EV0;
    // H of the inverse-L proposal: cached, or integrated by pot_flf_kernel for the cold particles
    float Hflf = MODE == kModeMJHMC ? a.Hflf_in[p] : 0.f;
    if (!(Hflf == Hflf)) Hflf = a.Hwork[p];
    Tile<NB> x, v, g;
    tile_load<NB>(a.X_in, p, w, h, x);
    tile_load<NB>(a.V_in, p, w, h, v);
    tile_load<NB>(a.G_in, p, w, h, g);
    float EXL = 0.f;
    pot_trajectory<NB>(mdl, ar, sh, w, c, h, lane, x, v, g, a.L, a.eps, a.chalf, &EXL);
    const float EVL = pot_kinetic<NB>(sh, w, c, h, v);
    const float HL = EXL + EVL;

    // rates / acceptance, waiting times, first minimum: lanes 0..31 of wave 0, one particle each
    if (w == 0 && h == 0) {
      const uint32_t pid = (uint32_t)(a.first_pid + (alive ? p : 0));
      double best = 0.0;
      bool bad = false, gate = false;
      int k;
      if constexpr (MODE == kModeMJHMC)
        k = dense_decide<REPLAY>(H0, HL, Hflf, a.p_r, pid, alive ? p : 0, a.N, a.rexp, a.key, best, bad);
      else if constexpr (MODE == kModeCT)
        k = dense_decide_ct<REPLAY>(H0, HL, a.p_r, pid, alive ? p : 0, a.N, a.rexp, a.key, best, bad);
      else
        k = dense_control<REPLAY>(H0, HL, a.p_r, a.p_flip, pid, alive ? p : 0, a.N, a.runif, a.key, gate);
      any_bad |= (bad && alive);
      if constexpr (MODE == kModeMJHMC) append_cold(a.next_list, a.next_count, alive && k != 0, p);
      sh.move[c] = k | (gate ? 4 : 0);
      a.dwell[p] = best;
      a.dwell_ring[p] = best;
      a.trans[p] = (uint8_t)k;
      if (alive) {
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
      }
      // scalars of the successors that keep or take whole states; a refreshed kinetic energy is filled in below
      const bool took_L = MODE == kModeControl ? (k & 1) : (k == 0);
      a.EX_out[p] = took_L ? EXL : EX0;
      a.EV_out[p] = took_L ? EVL : EV0;
      a.Hflf_out[p] = (MODE == kModeMJHMC && k == 0) ? H0 : __builtin_nanf("");
    }
    __syncthreads();
    const int mv = sh.move[c];
    const int k = mv & 3;
    bool refresh;  // this column's momentum is redrawn (HMCState.R)
    if constexpr (MODE == kModeControl) {
      if (!(k & 1)) {  // rejected: back to the pre-move state
        tile_load<NB>(a.X_in, p, w, h, x);
        tile_load<NB>(a.G_in, p, w, h, g);
        tile_load<NB>(a.V_in, p, w, h, v);
      } else {  // accepted L F: flip
#pragma unroll
        for (int r = 0; r < NB; ++r) v.b[r] = -v.b[r];
      }
      if (k & 2) {
#pragma unroll
        for (int r = 0; r < NB; ++r) v.b[r] = -v.b[r];
      }
      refresh = (mv & 4) != 0;  // batch-wide (markov_jump_hmc.py:138-141)
    } else {
      const bool keep_L = (k == 0);
      if (!keep_L) {  // F / R keep the position (and its gradient)
        tile_load<NB>(a.X_in, p, w, h, x);
        tile_load<NB>(a.G_in, p, w, h, g);
        tile_load<NB>(a.V_in, p, w, h, v);
      }
      if ((MODE == kModeCT && k == 0) || k == 1) {  // CT's FL move ends with a flip (:258,278); F flips
#pragma unroll
        for (int r = 0; r < NB; ++r) v.b[r] = -v.b[r];
      }
      refresh = (k == 2);
    }
    const bool tile_refreshes = __ballot(refresh) != 0ull;
    if constexpr (REPLAY) {
      if (refresh) {  // HMCState.R (hmc_state.py:121-129) with the recorded normals
        Tile<NB> z;
        tile_load<NB>(a.noise, alive ? p : 0, w, h, z);
#pragma unroll
        for (int r = 0; r < NB; ++r) v.b[r] = v.b[r] * a.r_keep + z.b[r] * a.r_mix;
      }
    } else {
      // column by column (the set is the same in every wave: it comes from sh.move), the whole workgroup drawing
      unsigned cols = (unsigned)(__ballot(refresh) & 0xFFFFFFFFull);
      while (cols) {
        const int c0 = __ffs((int)cols) - 1;
        cols &= cols - 1;
        const int64_t p0 = tile * kP + c0;
        column_normals<NB, float>(a.key, (uint32_t)(a.first_pid + (p0 < a.N ? p0 : 0)), a.D, sh.zn);
        __syncthreads();
        if (c == c0) {
          using V = typename VecN<NB>::type;
          const float* zrow = sh.zn + 32 * NB * w;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const V z = *reinterpret_cast<const V*>(zrow + NB * acc_row(q, h));
#pragma unroll
            for (int r = 0; r < NB; ++r) v.b[r][q] = v.b[r][q] * a.r_keep + vget<NB>(z, r) * a.r_mix;
          }
        }
        __syncthreads();
      }
    }
    if (tile_refreshes) {  // all waves take part in the reduction; only refreshed columns use the result
      const float evr = pot_kinetic<NB>(sh, w, c, h, v);
      if (refresh && w == 0 && h == 0) a.EV_out[p] = evr;
    }
    tile_store<NB>(a.X_out, p, w, h, x);
    tile_store<NB>(a.V_out, p, w, h, v);
    tile_store<NB>(a.G_out, p, w, h, g);
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

// ---------------------------------------------------------------------------------------------------
// HMCState.leapfrog / HMCState.L on caller-supplied states (hmc_state.py:86-100; figures/poe_fig.py:59 assigns and
// integrates states of a ProductOfT sampler): dE/dX at the start point, L steps, energies of the end point.
// ---------------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256, 1) void pot_leap_kernel(const PotLeapArgs a, const PotModel mdl) {
  __shared__ Shared<NB> sh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  AReg<NB> ar;
  areg_load<NB>(mdl, w, c, h, ar);
  stage_bias<NB>(mdl, sh);
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t p = tile * kP + c;
    Tile<NB> x, v, g;
    tile_load<NB>(a.X, p, w, h, x);
    tile_load<NB>(a.V, p, w, h, v);
    float ex = 0.f;
    pot_gradient<NB>(mdl, ar, sh, w, c, h, lane, x, g, true, &ex);
    __syncthreads();
    pot_trajectory<NB>(mdl, ar, sh, w, c, h, lane, x, v, g, a.L, a.eps, a.chalf, &ex);
    const float ev = pot_kinetic<NB>(sh, w, c, h, v);
    tile_store<NB>(a.X_out, p, w, h, x);
    tile_store<NB>(a.V_out, p, w, h, v);
    if (a.G) tile_store<NB>(a.G, p, w, h, g);
    if (w == 0 && h == 0) {
      if (a.EX) a.EX[p] = ex;
      if (a.EV) a.EV[p] = ev;
    }
    __syncthreads();
  }
}

#ifdef POT_STAMPS
}  // namespace mjhmc
extern "C" int mjhmc_pot_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mjhmc::g_pot_stamp), sizeof(mjhmc::g_pot_stamp));
}
namespace mjhmc {
#endif

static int resident_cus() {
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return std::max(1, cus);
}

// ---------------------------------------------------------------------------------------------------
// ndims == nbasis > 512: blocked evaluation (multi-pass path only).  C[:, cblk] (+)= M_block^T-form GEMM of B[:, bblk]:
// one launch per 512 x 512 block of the matrix, the tile kernels' own GEMM (gemm_dim<4>) on a tile of 32 rows.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_load_ld(const float* base, int64_t ld, int64_t p, int w, int h, Tile<4>& t) {
  const float* row = base + (size_t)p * ld + 128 * w;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 4 * acc_row(q, h));
#pragma unroll
    for (int r = 0; r < 4; ++r) t.b[r][q] = v[r];
  }
}
__device__ __forceinline__ void tile_store_ld(float* base, int64_t ld, int64_t p, int w, int h, const Tile<4>& t) {
  float* row = base + (size_t)p * ld + 128 * w;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = t.b[r][q];
    *reinterpret_cast<f32x4*>(row + 4 * acc_row(q, h)) = v;
  }
}

// C tile (32 rows x this block's 512 columns) = [init vector | C tile] + sum over the block's 512 k-rows
__global__ __launch_bounds__(256, 1) void pot_block_gemm_kernel(const float* __restrict__ M, const float* __restrict__ B, float* C,
                                                                int64_t ld, const float* __restrict__ initvec, int accumulate,
                                                                int64_t ntiles) {
  __shared__ Shared<4> sh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  typename VecN<4>::type a0[16];
  a_chunk_load<4>(M + (size_t)(16 * h) * 512 + 128 * w + 4 * c, 0, a0);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t p = tile * kP + c;
    Tile<4> x, acc;
    tile_load_ld(B, ld, p, w, h, x);
    publish<4>(sh.pub[0][w], lane, x);
    if (accumulate) tile_load_ld(C, ld, p, w, h, acc);
    else if (initvec) rowvec_load<4>(initvec, w, h, acc);
    else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc.b[r][q] = 0.f;
    }
    __syncthreads();
    gemm_dim<4>(M, M, sh.pub[0], w, c, h, lane, a0, acc);   // (a0 leaves holding chunk 0 of the same block: the next tile's)
    tile_store_ld(C, ld, p, w, h, acc);
    __syncthreads();
  }
}

// E = sum_j alpha_j log(1 + u_j^2) per row (distributions.py:430-432); u <- phi(u) = u / (1 + u^2) in place
__global__ void pot_big_phi_kernel(float* __restrict__ U, float* __restrict__ E, const float* __restrict__ alpha, int dim,
                                   int64_t rows) {
  const int64_t r = blockIdx.x;
  if (r >= rows) return;
  float* u = U + (size_t)r * dim;
  float s = 0.f;
  for (int j = threadIdx.x; j < dim; j += 64) {
    const float uu = u[j];
    if (E) s += alpha[j] * logf(1.0f + uu * uu);
    u[j] = uu * __builtin_amdgcn_rcpf(1.0f + uu * uu);
  }
  if (E) {
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) E[r] = s;
  }
}

void pot_big_eval(const PotBigModel& m, const float* X32, float* G32, float* E32, float* U, int64_t rows_pad, hipStream_t st) {
  const int nb = m.dim / 512;
  const int64_t ntiles = rows_pad / 32;
  const unsigned grid = (unsigned)std::min<int64_t>(ntiles, resident_cus());
  const size_t blk = (size_t)512 * 512;
  for (int jb = 0; jb < nb; ++jb)      // u[:, jb] = cb[jb] + sum_db W1[db, jb]^T x[:, db]
    for (int db = 0; db < nb; ++db)
      hipLaunchKernelGGL(pot_block_gemm_kernel, dim3(grid), dim3(256), 0, st, m.W1b + ((size_t)db * nb + jb) * blk,
                         X32 + (size_t)db * 512, U + (size_t)jb * 512, (int64_t)m.dim, m.cb + (size_t)jb * 512, db > 0 ? 1 : 0, ntiles);
  hipLaunchKernelGGL(pot_big_phi_kernel, dim3((unsigned)rows_pad), dim3(64), 0, st, U, E32, m.alpha, m.dim, rows_pad);
  if (!G32) return;
  for (int db = 0; db < nb; ++db)      // dE/dx[:, db] = sum_jb W2T[jb, db]^T phi(u)[:, jb]
    for (int jb = 0; jb < nb; ++jb)
      hipLaunchKernelGGL(pot_block_gemm_kernel, dim3(grid), dim3(256), 0, st, m.W2Tb + ((size_t)jb * nb + db) * blk,
                         (const float*)U + (size_t)jb * 512, G32 + (size_t)db * 512, (int64_t)m.dim, (const float*)nullptr, jb > 0 ? 1 : 0,
                         ntiles);
}

template <int NB, int MODE>
static void launch_jump_mode(const PotJumpArgs& a, const PotModel& mdl, unsigned grid, hipStream_t st) {
  const bool replay = MODE == kModeControl ? (a.runif && a.noise) : (a.rexp && a.noise);
  if (replay) hipLaunchKernelGGL((pot_jump_kernel<NB, true, MODE>), dim3(grid), dim3(256), 0, st, a, mdl);
  else hipLaunchKernelGGL((pot_jump_kernel<NB, false, MODE>), dim3(grid), dim3(256), 0, st, a, mdl);
}

template <int NB>
static void launch_jump_nb(const PotJumpArgs& a, const PotModel& mdl, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, resident_cus());
  if (a.mode == kModeMJHMC) {  // only MJHMC has the inverse-L proposal and its cache
    if (a.iter == 0) {  // first iteration of a call: both counters cleared (they are adjacent), the list from a scan
      (void)hipMemsetAsync(a.cold_count < a.next_count ? a.cold_count : a.next_count, 0, 2 * sizeof(int), st);
      hipLaunchKernelGGL(pot_cold_list_kernel, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, st, a.Hflf_in, a.N,
                         a.cold_list, a.cold_count, (const Control*)a.ctl);
    }
    hipLaunchKernelGGL(pot_flf_kernel<NB>, dim3(grid), dim3(256), 0, st, a, mdl);
    launch_jump_mode<NB, kModeMJHMC>(a, mdl, grid, st);
  } else if (a.mode == kModeCT) {
    launch_jump_mode<NB, kModeCT>(a, mdl, grid, st);
  } else {
    launch_jump_mode<NB, kModeControl>(a, mdl, grid, st);
  }
}

void pot_launch_jump(const PotJumpArgs& a, const PotModel& mdl, hipStream_t st) {
  if (mdl.dim == 128) launch_jump_nb<1>(a, mdl, st);
  else if (mdl.dim == 256) launch_jump_nb<2>(a, mdl, st);
  else launch_jump_nb<4>(a, mdl, st);
}

void pot_launch_leap(const PotLeapArgs& a, const PotModel& mdl, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, resident_cus());
  if (mdl.dim == 128) hipLaunchKernelGGL(pot_leap_kernel<1>, dim3(grid), dim3(256), 0, st, a, mdl);
  else if (mdl.dim == 256) hipLaunchKernelGGL(pot_leap_kernel<2>, dim3(grid), dim3(256), 0, st, a, mdl);
  else hipLaunchKernelGGL(pot_leap_kernel<4>, dim3(grid), dim3(256), 0, st, a, mdl);
}

void pot_launch_eval(const PotEvalArgs& a, const PotModel& mdl, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, resident_cus());
  if (mdl.dim == 128) hipLaunchKernelGGL(pot_eval_kernel<1>, dim3(grid), dim3(256), 0, st, a, mdl);
  else if (mdl.dim == 256) hipLaunchKernelGGL(pot_eval_kernel<2>, dim3(grid), dim3(256), 0, st, a, mdl);
  else hipLaunchKernelGGL(pot_eval_kernel<4>, dim3(grid), dim3(256), 0, st, a, mdl);
}

}  // namespace mjhmc
