// Argument blocks of the ProductOfT MFMA kernels (dense_pot.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elementwise.hpp"  // Control, RngKey, philox

namespace mjhmc {

constexpr int kPotDim = 512;  // largest ndims == nbasis of the register-resident tile kernels (rows are zero padded to 128, 256 or 512);
                              // beyond it: blocked evaluation on the multi-pass path (PotBigModel, pot_big_eval)

// device-resident model, float32, padded to 512 x 512
struct PotModel {
  const float* W1;     // [d][j] = W[d][j] / nu_j              (first GEMM:  u = W1^T x + cb)
  const float* W2T;    // [j][d] = W[d][j] * (nu_j + 1) / nu_j  (second GEMM: dE/dx = W2T^T phi(u))
  const float* cb;     // [j]    = b_j / nu_j
  const float* alpha;  // [j]    = (nu_j + 1) / 2   (0 for padded experts)
  int dim;             // padded dimension: 128, 256 or 512
  int ndims;           // true ndims == nbasis (rows at or beyond it are zero padding)
};

struct PotJumpArgs {
  const float* X_in;
  const float* V_in;
  const float* G_in;   // dE/dX at X_in (HMCState.dEdX, hmc_state.py:36-39)
  float* X_out;
  float* V_out;
  float* G_out;
  const float* EX_in;
  const float* EV_in;
  const float* Hflf_in;
  float* Hwork;        // [Npad] H of the inverse-L proposal, where this iteration integrates it (the listed particles)
  int* cold_list;      // [Npad] compacted indices of the particles whose inverse-L proposal must be integrated
  int* cold_count;
  int* next_list;       // the NEXT iteration's list and counter, filled by this iteration's jump and fix kernels
  int* next_count;
  int* zero_count;      // the counter of the list two iterations back (consumed): cleared by the jump kernel for its next use
  int rescan;           // host side: build this iteration's list by a scan even if it is not the call's first (test build, MJHMC_NO_FSPEC)
  const float* Hspec_in;   // H of the L proposal of a particle that then moved by F, NaN otherwise (see dense_pot.hip: F-movers)
  float* Hspec_out;
  float* EX_out;
  float* EV_out;
  float* Hflf_out;
  double* dwell;
  double* dwell_ring;
  uint8_t* trans;
  const float* noise;   // replay normals [N][512] or nullptr
  const double* rexp;   // replay unit exponentials [3][N] or nullptr
  const double* runif;  // replay uniforms of the discrete-time samplers [2N+1] (accept, flip, R gate) or nullptr
  Control* ctl;
  unsigned long long* stats;
  int64_t N, Npad, ntiles, first_pid;
  int D, L, iter;
  int mode;             // kModeMJHMC / kModeControl / kModeCT
  float eps, chalf, r_keep, r_mix;
  double p_r, p_flip;
  RngKey key;
};

// The same iteration in the reference's arithmetic (dense_pot64.hip): float64 state rows around the float32 force
struct Pot64JumpArgs {
  const double* X_in;
  const double* V_in;
  const double* G_in;   // dE/dX at X_in: float32 values widened (hmc_state.py:52-53)
  double* X_out;        // also the working rows of the trajectory
  double* V_out;
  double* G_out;
  const double* EX_in;
  const double* EV_in;
  const double* Hflf_in;
  double* Hwork;
  int* cold_list;
  int* cold_count;
  int* next_list;       // the NEXT iteration's list and counter, filled by this iteration's jump and fix kernels
  int* next_count;
  int* zero_count;
  int rescan;
  const double* Hspec_in;
  double* Hspec_out;
  double* EX_out;
  double* EV_out;
  double* Hflf_out;
  double* dwell;
  double* dwell_ring;
  uint8_t* trans;
  const double* noise;
  const double* rexp;
  const double* runif;
  double* scratch;      // [workgroups][2][32][dim] working rows of the inverse-L pass (pot64_scratch_workgroups())
  Control* ctl;
  unsigned long long* stats;
  int64_t N, Npad, ntiles, first_pid;
  int D, L, iter;       // L >= 1
  int mode;
  double eps, chalf, r_keep, r_mix;
  double p_r, p_flip;
  RngKey key;
};

// stand-alone leapfrog operator on caller-supplied states (HMCState.leapfrog / L, hmc_state.py:86-100)
struct PotLeapArgs {
  const float* X;
  const float* V;
  float* X_out;
  float* V_out;
  float* G;        // dE/dX at the end point, or nullptr
  float* EX;       // [n] or nullptr
  float* EV;
  int64_t N, ntiles;
  int D, L;
  float eps, chalf;
};

struct PotEvalArgs {
  const float* X;
  float* G;
  float* E;
  float* EV;
  const float* V;
  float* V_gen;
  int64_t N, ntiles, first_pid;
  int D;
  RngKey key;
};

// Rates, waiting times and first minimum of ONE particle, evaluated serially by one lane (the dense
// kernels run it on 32 lanes of one wave, one particle each; ~1e3 instructions against ~1e6 cycles of
// trajectory).  markov_jump_hmc.py:366-396, utils.py:15-49.  Returns k (0 L, 1 F, 2 R).
template <bool REPLAY>
__device__ __forceinline__ int dense_decide(float H0, float HL, float Hflf, double p_r, uint32_t pid, int64_t p,
                                            int64_t N, const double* rexp, const RngKey& key, double& dwell, bool& bad) {
  const double l_rate = sqrt(exp((double)(H0 - HL)));
  const double flf_rate = sqrt(exp((double)(H0 - Hflf)));
  const double mn = (flf_rate != flf_rate || l_rate != l_rate) ? __builtin_nan("") : fmin(flf_rate, l_rate);
  const double f_rate = flf_rate - mn;
  const double r_rate = p_r;
  double eL, eF, eR;
  if constexpr (REPLAY) {
    eL = rexp[p];
    eF = rexp[N + p];
    eR = rexp[2 * N + p];
  } else {
    const u32x4 wq = philox4x32_10(pid, key.tick_lo, key.tick_hi, kSlotExpLF, key.k0, key.k1);
    const u32x4 qq = philox4x32_10(pid, key.tick_lo, key.tick_hi, kSlotExpR, key.k0, key.k1);
    eL = -log(u53(wq.w0, wq.w1));
    eF = -log(u53(wq.w2, wq.w3));
    eR = -log(u53(qq.w0, qq.w1));
  }
  bad = !(isfinite(l_rate) && isfinite(f_rate) && isfinite(r_rate));
  const double dL = l_rate == 0.0 ? __builtin_huge_val() : (1.0 / l_rate) * eL;
  const double dF = f_rate == 0.0 ? __builtin_huge_val() : (1.0 / f_rate) * eF;
  const double dR = r_rate == 0.0 ? __builtin_huge_val() : (1.0 / r_rate) * eR;
  int k = 0;
  double best = dL;
  if (!(best != best) && (dF < best || dF != dF)) {
    k = 1;
    best = dF;
  }
  if (!(best != best) && (dR < best || dR != dR)) {
    k = 2;
    best = dR;
  }
  dwell = best;
  return k;
}

// ContinuousTimeHMC (markov_jump_hmc.py:251-290): clocks FL (rate sqrt(exp(H0 - H_fl))), F (rate 1), R (rate p_r);
// min_idx is called with [f, fl, r], so ties go F, FL, R.  Returns k: 0 = FL, 1 = F, 2 = R.
template <bool REPLAY>
__device__ __forceinline__ int dense_decide_ct(float H0, float HL, double p_r, uint32_t pid, int64_t p, int64_t N,
                                               const double* rexp, const RngKey& key, double& dwell, bool& bad) {
  const double fl_rate = sqrt(exp((double)(H0 - HL)));
  const double r_rate = p_r;
  double eFL, eF, eR;
  if constexpr (REPLAY) {
    eFL = rexp[p];
    eF = rexp[N + p];
    eR = rexp[2 * N + p];
  } else {
    const u32x4 wq = philox4x32_10(pid, key.tick_lo, key.tick_hi, kSlotExpLF, key.k0, key.k1);
    const u32x4 qq = philox4x32_10(pid, key.tick_lo, key.tick_hi, kSlotExpR, key.k0, key.k1);
    eFL = -log(u53(wq.w0, wq.w1));
    eF = -log(u53(wq.w2, wq.w3));
    eR = -log(u53(qq.w0, qq.w1));
  }
  bad = !(isfinite(fl_rate) && isfinite(r_rate));
  const double dFL = fl_rate == 0.0 ? __builtin_huge_val() : (1.0 / fl_rate) * eFL;
  const double dF = eF;  // rate 1
  const double dR = r_rate == 0.0 ? __builtin_huge_val() : (1.0 / r_rate) * eR;
  int kk = 0;  // rows f, fl, r (markov_jump_hmc.py:271)
  double best = dF;
  if (!(best != best) && (dFL < best || dFL != dFL)) {
    kk = 1;
    best = dFL;
  }
  if (!(best != best) && (dR < best || dR != dR)) {
    kk = 2;
    best = dR;
  }
  dwell = best;
  return kk == 0 ? 1 : (kk == 1 ? 0 : 2);
}

// Discrete-time control samplers (markov_jump_hmc.py:116-148): accept the L F proposal with min(1, exp(H0 - H1)) (a NaN
// difference accepts, as `Ediff < 0` is False, :112-113), flip with probability p_flip; `gate`: the batch-wide
// momentum refresh fires this iteration.  Returns k = accepted | flipped << 1.
template <bool REPLAY>
__device__ __forceinline__ int dense_control(float H0, float HL, double p_r, double p_flip, uint32_t pid, int64_t p,
                                             int64_t N, const double* runif, const RngKey& key, bool& gate) {
  double uacc, uflip, ugate;
  if constexpr (REPLAY) {
    uacc = runif[p];
    uflip = runif[N + p];
    ugate = runif[2 * N];
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
  const bool flip = uflip < p_flip;
  gate = ugate < p_r;
  return (accept ? 1 : 0) | (flip ? 2 : 0);
}

// float32 Box-Muller pair from the same Philox words as normal_pair (the float64 version costs ~4x
// the instructions; for float32 / bfloat16 state its extra digits are rounded away anyway)
__device__ __forceinline__ void normal_pair_f32(const RngKey& k, uint32_t pid, uint32_t pair, float& z0, float& z1) {
  const u32x4 w = philox4x32_10(pid, k.tick_lo, k.tick_hi, pair, k.k0, k.k1);
  const float u1 = (float)u53(w.w0, w.w1);
  const float u2 = (float)u53(w.w2, w.w3);
  const float r = sqrtf(-2.0f * logf(u1 > 0.f ? u1 : 1e-38f));
  float s, c;
  sincospif(2.0f * u2, &s, &c);
  z0 = r * c;
  z1 = r * s;
}

// ndims == nbasis > 512 (the reference takes any square size, distributions.py:379-406): the pre-scaled matrices as
// 512 x 512 BLOCKS, each laid out like a whole 512-dim model's matrix, so that the tile kernels' GEMM runs on them as is
struct PotBigModel {
  const float* W1b;    // [db][jb][512][512]: block (db, jb) of W[d][j] / nu_j
  const float* W2Tb;   // [jb][db][512][512]: block (jb, db) of (W[d][j] (nu_j + 1) / nu_j)^T
  const float* cb;     // [dim]
  const float* alpha;  // [dim]
  int dim;             // ndims rounded up to a multiple of 512
  int ndims;
};
// E (or nullptr) and dE/dX (or nullptr) of rows X32 [rows_pad][dim] (float32, rows_pad a multiple of 32); U: scratch of
// the same shape.  2 (dim / 512)^2 block-GEMM launches + one elementwise pass.
void pot_big_eval(const PotBigModel& m, const float* X32, float* G32, float* E32, float* U, int64_t rows_pad, hipStream_t st);

void pot_launch_jump(const PotJumpArgs& a, const PotModel& mdl, hipStream_t st);
void pot_launch_eval(const PotEvalArgs& a, const PotModel& mdl, hipStream_t st);
void pot_launch_leap(const PotLeapArgs& a, const PotModel& mdl, hipStream_t st);
void pot64_launch_jump(const Pot64JumpArgs& a, const PotModel& mdl, hipStream_t st);
int pot64_scratch_workgroups();  // workgroups a pot64 launch may run: rows of Pot64JumpArgs::scratch to provide per launch

}  // namespace mjhmc
