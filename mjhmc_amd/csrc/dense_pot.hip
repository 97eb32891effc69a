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
// The inverse-L proposal F L F of MarkovJumpHMC (markov_jump_hmc.py:360-367, hmc_state.py:109-119) on the matrix cores.
//
// WHICH particles integrate it.  The reference integrates F L F for every particle whose cache is cold -- the F- and
// R-movers of the iteration before (markov_jump_hmc.py:409-411) -- and reads the result only through H().  For an
// F-MOVER that trajectory is one this kernel has just run: the state after F is (X, -V), so F L F (X, -V) = F L (X, V),
// and L (X, V) is the L proposal of the iteration in which the particle flipped -- the same start point, the same
// stored dE/dX, the same operations in the same order: the same bits, and H() does not see the final F.  (The
// reference's authors knew: `# self.state.cache_flf_state(f_idx, l_state.F())` stands commented out at
// markov_jump_hmc.py:401.)  So the jump kernel hands an F-mover's H(L proposal) on in `Hspec` and the next iteration
// uses it instead of integrating: only R-movers (a fresh momentum) are integrated.  The cache stays COLD as far as anyone
// can see -- H_flf is NaN, cache_active False, the evaluation counters count the particle as the reference does
// (stats[3]: all cold particles in its low half, what was actually integrated in its high half) -- and `Hspec` is dropped (NaN) by anything
// that could make it stale: new hyper-parameters, a state write, reset_flf_cache, restore (api.hip: drop_spec).
// tests/test_gpu_dense_parity.py::test_f_mover_shortcut_is_bit_identical runs both ways.
//
// WHEN.  A tile of 32 particles costs the same whether 1 or 32 of them need the trajectory, so the listed particles are
// gathered into dense tiles; and those tiles are items of the SAME launch as the iteration's forward tiles (items
// [0, nft) = inverse-L tiles of the list, [nft, nft + ntiles) = forward tiles): they run beside each other instead of
// one kernel after the other -- a list of 20 tiles used to hold the chip for a whole trajectory time on 20 CUs, which
// at N / 8 particles was a third of the iteration.  A listed particle's decision needs both of its trajectories, so the
// jump kernel leaves it PENDING (its L proposal written as if taken, no bookkeeping) and pot_fix_kernel, a tile kernel
// over the same list without any trajectory, decides it afterwards and puts the pre-move state back where the move
// is not L.  Same device functions, same per-column reductions: a particle's results do not depend on which kernel
// finished it.
//
// The list of the FIRST iteration of a mjhmc_iterate call comes from a scan (this kernel, once per call); every later
// one is appended by the jump and fix kernels of the iteration before it (append_cold): no per-iteration memset, no
// per-iteration list kernel -- a 4-byte memset behind a persistent grid waited for a workgroup to exit.  Counters
// rotate over three slots: iteration i reads slot i % 3, appends to (i + 1) % 3 and clears (i + 2) % 3.
// ---------------------------------------------------------------------------------------------------
__global__ void pot_cold_list_kernel(const float* __restrict__ Hflf_in, const float* __restrict__ Hspec_in, int64_t N,
                                     int* __restrict__ list, int* __restrict__ count, const Control* ctl) {
  if (ctl->failed) return;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float hc = p < N ? Hflf_in[p] : 0.f, hs = p < N ? Hspec_in[p] : 0.f;
  append_cold(list, count, (p < N) && !(hc == hc) && !(hs == hs), p);
}

// The successor's rows once the moves of a tile's columns stand in sh.move.  FIX = false (jump kernel): x, v, g hold the
// end point of L.  FIX = true (pot_fix_kernel): columns that keep the end point are finished already (their rows hold
// it); only the others are touched.
template <int NB, bool REPLAY, int MODE, bool FIX, class SH>
__device__ __forceinline__ void pot_finish(const PotJumpArgs& a, SH& sh, int64_t p, bool alive, int w, int c, int h,
                                           Tile<NB>& x, Tile<NB>& v, Tile<NB>& g) {
  const int mv = sh.move[c];
  const int k = mv & 3;
  bool refresh;  // this column's momentum is redrawn (HMCState.R)
  bool touch = true;
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
    if constexpr (FIX) touch = !keep_L;
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
      const int64_t p0 = __shfl((long long)p, c0);
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
  if (touch) {
    tile_store<NB>(a.X_out, p, w, h, x);
    tile_store<NB>(a.V_out, p, w, h, v);
    tile_store<NB>(a.G_out, p, w, h, g);
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
  // MJHMC: the inverse-L tiles of this iteration's list are the first items of the launch
  const int ncold = MODE == kModeMJHMC ? *a.cold_count : 0;
  const int64_t nft = (ncold + kP - 1) / kP;
  if ((int64_t)blockIdx.x >= nft + a.ntiles) return;
  if (MODE == kModeMJHMC && blockIdx.x == 0 && threadIdx.x == 0) {
    *a.zero_count = 0;   // the list two iterations back is consumed: its counter is free for the next iteration's appends
    if (ncold) atomicAdd(&a.stats[3], (unsigned long long)ncold << 32);   // integrated here: the high half of the cold tally
  }
  unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0;  // tallies (meaning per mode: fill_iter_stats in api.hip)
  bool any_bad = false;
  AReg<NB> ar;
  areg_load<NB>(mdl, w, c, h, ar);
  stage_bias<NB>(mdl, sh);
  for (int64_t item = blockIdx.x; item < nft + a.ntiles; item += gridDim.x) {
    const bool inverse = item < nft;   // (uniform over the workgroup)
    int64_t p;
    if (inverse) {
      const int64_t slot = item * kP + c;
      p = a.cold_list[slot < ncold ? slot : ncold - 1];  // pad the last tile with a repeat
    } else {
      p = (item - nft) * kP + c;
    }
    const bool alive = p < a.N;
    Tile<NB> x, v, g;
    tile_load<NB>(a.X_in, p, w, h, x);
    tile_load<NB>(a.V_in, p, w, h, v);
    tile_load<NB>(a.G_in, p, w, h, g);
    if (inverse) {
#pragma unroll
      for (int r = 0; r < NB; ++r) v.b[r] = -v.b[r];
    }
    float EXL = 0.f;
    pot_trajectory<NB>(mdl, ar, sh, w, c, h, lane, x, v, g, a.L, a.eps, a.chalf, &EXL);
    const float EVL = pot_kinetic<NB>(sh, w, c, h, v);
    const float HL = EXL + EVL;
    if (inverse) {
      if (w == 0 && h == 0) a.Hwork[p] = HL;
      __syncthreads();
      continue;
    }

    // rates / acceptance, waiting times, first minimum: lanes 0..31 of wave 0, one particle each
    if (w == 0 && h == 0) {
      const uint32_t pid = (uint32_t)(a.first_pid + (alive ? p : 0));
      const float EX0 = a.EX_in[p], EV0 = a.EV_in[p];
      const float H0 = EX0 + EV0;
      // H of the inverse-L proposal: cached; or the L proposal of the iteration in which the particle flipped; or being
      // integrated by an inverse-L item of this very launch -- then the particle is left pending for pot_fix_kernel
      float Hflf = MODE == kModeMJHMC ? a.Hflf_in[p] : 0.f;
      const bool cold = !(Hflf == Hflf);
      bool pending = false;
      if (cold) {
        Hflf = a.Hspec_in[p];
        pending = !(Hflf == Hflf);
      }
      double best = 0.0;
      bool bad = false, gate = false;
      int k = 0;
      if (!pending) {
        if constexpr (MODE == kModeMJHMC)
          k = dense_decide<REPLAY>(H0, HL, Hflf, a.p_r, pid, alive ? p : 0, a.N, a.rexp, a.key, best, bad);
        else if constexpr (MODE == kModeCT)
          k = dense_decide_ct<REPLAY>(H0, HL, a.p_r, pid, alive ? p : 0, a.N, a.rexp, a.key, best, bad);
        else
          k = dense_control<REPLAY>(H0, HL, a.p_r, a.p_flip, pid, alive ? p : 0, a.N, a.runif, a.key, gate);
        any_bad |= (bad && alive);
        // every move but L clears the cache; of those only the R-movers need their inverse-L proposal integrated
        if constexpr (MODE == kModeMJHMC) append_cold(a.next_list, a.next_count, alive && k == 2, p);
        a.dwell[p] = best;
        a.dwell_ring[p] = best;
        a.trans[p] = (uint8_t)k;
      }
      sh.move[c] = k | (gate ? 4 : 0);
      if (alive) {
        if constexpr (MODE == kModeControl) {  // l_count, f_count, R applied, fl_count (markov_jump_hmc.py:143-148)
          n0 += (k == 3);
          n1 += (k == 2);
          n2 += gate ? 1u : 0u;
          n3 += (k == 1);
        } else if (!pending) {
          n0 += (k == 0);
          n1 += (k == 1);
          n2 += (k == 2);
        }
        if constexpr (MODE == kModeMJHMC) n3 += cold;   // the reference integrates F L F for every one of these
      }
      // scalars of the successors that keep or take whole states; a refreshed kinetic energy is filled in below
      const bool took_L = MODE == kModeControl ? (k & 1) : (k == 0);
      a.EX_out[p] = took_L ? EXL : EX0;
      a.EV_out[p] = took_L ? EVL : EV0;
      if (!pending) {
        a.Hflf_out[p] = (MODE == kModeMJHMC && k == 0) ? H0 : __builtin_nanf("");
        if constexpr (MODE == kModeMJHMC) a.Hspec_out[p] = (k == 1) ? HL : __builtin_nanf("");
      }
    }
    __syncthreads();
    pot_finish<NB, REPLAY, MODE, false>(a, sh, p, alive, w, c, h, x, v, g);
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

// The particles the jump kernel left pending (this iteration's list): both trajectories are done now -- H of the
// inverse-L proposal in Hwork, the L proposal in the output rows with its energies in EX_out / EV_out -- so decide, and
// where the move is not L put the pre-move position and dE/dX back and flip / redraw the momentum.
template <int NB, bool REPLAY>
__global__ __launch_bounds__(256) void pot_fix_kernel(const PotJumpArgs a) {
  __shared__ FinishShared<NB> sh;
  if (a.ctl->failed) return;
  const int ncold = *a.cold_count;
  if ((int64_t)blockIdx.x * kP >= ncold) return;
  const int w = threadIdx.x >> 6, c = threadIdx.x & 31, h = (threadIdx.x & 63) >> 5;
  unsigned n0 = 0, n1 = 0, n2 = 0;
  bool any_bad = false;
  for (int64_t tile = blockIdx.x; tile * kP < ncold; tile += gridDim.x) {
    const int64_t slot = tile * kP + c;
    const bool valid = slot < ncold;    // the last tile repeats an entry: those columns do nothing
    const int64_t p = a.cold_list[valid ? slot : ncold - 1];
    if (w == 0 && h == 0) {
      const uint32_t pid = (uint32_t)(a.first_pid + p);
      const float EX0 = a.EX_in[p], EV0 = a.EV_in[p];
      const float H0 = EX0 + EV0;
      const float HL = a.EX_out[p] + a.EV_out[p];
      double best = 0.0;
      bool bad = false;
      const int k = dense_decide<REPLAY>(H0, HL, a.Hwork[p], a.p_r, pid, p, a.N, a.rexp, a.key, best, bad);
      sh.move[c] = valid ? k : 0;
      if (valid) {
        any_bad |= bad;
        append_cold(a.next_list, a.next_count, k == 2, p);
        a.dwell[p] = best;
        a.dwell_ring[p] = best;
        a.trans[p] = (uint8_t)k;
        n0 += (k == 0);
        n1 += (k == 1);
        n2 += (k == 2);
        if (k != 0) {
          a.EX_out[p] = EX0;
          a.EV_out[p] = EV0;   // (an R-mover's: filled in by pot_finish)
        }
        a.Hflf_out[p] = k == 0 ? H0 : __builtin_nanf("");
        a.Hspec_out[p] = k == 1 ? HL : __builtin_nanf("");
      }
    }
    __syncthreads();
    Tile<NB> x, v, g;
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) v.b[r][q] = 0.f;
    pot_finish<NB, REPLAY, kModeMJHMC, true>(a, sh, p, true, w, c, h, x, v, g);
    __syncthreads();
  }
  if (any_bad) {
    a.ctl->failed = 1;
    a.ctl->failed_iter = a.iter;
  }
  __shared__ unsigned tally[3];
  if (threadIdx.x < 3) tally[threadIdx.x] = 0;
  __syncthreads();
  if (n0) atomicAdd(&tally[0], n0);
  if (n1) atomicAdd(&tally[1], n1);
  if (n2) atomicAdd(&tally[2], n2);
  __syncthreads();
  if (threadIdx.x < 3 && tally[threadIdx.x]) atomicAdd(&a.stats[threadIdx.x], (unsigned long long)tally[threadIdx.x]);
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
  const int cus = resident_cus();
  if (a.mode == kModeMJHMC) {  // only MJHMC has the inverse-L proposal and its cache
    if (a.iter == 0 || a.rescan) {  // first iteration of a call: the three counters cleared (they are adjacent), the list from a scan
      (void)hipMemsetAsync(std::min(a.cold_count, std::min(a.next_count, a.zero_count)), 0, 3 * sizeof(int), st);
      hipLaunchKernelGGL(pot_cold_list_kernel, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, st, a.Hflf_in, a.Hspec_in,
                         a.N, a.cold_list, a.cold_count, (const Control*)a.ctl);
    }
    // forward tiles + at most as many inverse-L tiles (workgroups without an item leave at once)
    const unsigned grid = (unsigned)std::min<int64_t>(2 * a.ntiles, cus);
    launch_jump_mode<NB, kModeMJHMC>(a, mdl, grid, st);
    const unsigned fgrid = (unsigned)std::min<int64_t>(a.ntiles, 4 * cus);
    if (a.rexp && a.noise) hipLaunchKernelGGL((pot_fix_kernel<NB, true>), dim3(fgrid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((pot_fix_kernel<NB, false>), dim3(fgrid), dim3(256), 0, st, a);
  } else {
    const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, cus);
    if (a.mode == kModeCT) launch_jump_mode<NB, kModeCT>(a, mdl, grid, st);
    else launch_jump_mode<NB, kModeControl>(a, mdl, grid, st);
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
