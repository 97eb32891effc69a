// Argument blocks of the SparseImageCode bf16-MFMA kernels (dense_sic.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dense_pot.hpp"  // dense_decide, normal_pair_f32, Control, RngKey

namespace mjhmc {

constexpr int kSicImg = 256;      // img_size
// n_coeffs: 1024 or 512, the dictionaries the reference accepts (tf_distributions.py:219)
inline bool sic_coeffs_supported(int nc) { return nc == 1024 || nc == 512; }

struct SicModel {
  const void* A1;   // (rounds 1-2: the dictionary in GEMM1's fragment order; the one-pass kernel reads A2 only)
  const void* A2;   // bf16 [16 k-steps][2 halves][nc coeffs][8]:         B[i][c] in GEMM2's fragment k-order
  const float* y;   // [n_patches][256] the patches
  float lambda;
  int cauchy;
  int P;            // n_patches: a particle is P consecutive nc-coefficient rows, one per patch (tf_distributions.py:228-229)
  float invP;       // the reconstruction error is the MEAN over patches (tf_distributions.py:259-260)
  int copies;       // identical copies of A1 / A2 laid out back to back
  int nc;           // n_coeffs: 1024 or 512
};
constexpr int kSicCopies = 1;

// ST = the type the state rows are STORED in: __bf16 (BASELINE.json configs[4]: "bf16 state / fp32 accumulate") or float
// (the reference's own: TensorFlow float32 placeholders, tf_distributions.py:89).  The matrix-core operands are bf16 either way.
template <typename ST>
struct SicJumpArgsT {
  const ST* X_in;
  const ST* V_in;
  ST* X_out;
  ST* V_out;
  const float* EX_in;
  const float* EV_in;
  const float* Hflf_in;
  float* Hwork;
  int* cold_list;
  int* cold_count;
  int* next_list;       // the NEXT iteration's list and counter, filled by this iteration's jump and fix kernels
  int* next_count;
  int* zero_count;      // (as PotJumpArgs)
  int rescan;
  const float* Hspec_in;
  float* Hspec_out;
  float* EX_out;
  float* EV_out;
  float* Hflf_out;
  double* dwell;
  double* dwell_ring;
  uint8_t* trans;
  const ST* noise;
  const double* rexp;
  const double* runif;  // replay uniforms of the discrete-time samplers [2N+1] or nullptr
  Control* ctl;
  unsigned long long* stats;
  int64_t N, Npad, ntiles, first_pid;
  int L, iter;
  int mode;             // kModeMJHMC / kModeControl / kModeCT
  float eps, chalf, r_keep, r_mix;
  double p_r, p_flip;
  RngKey key;
};

// stand-alone leapfrog operator on caller-supplied states (HMCState.leapfrog / L, hmc_state.py:86-100)
template <typename ST>
struct SicLeapArgsT {
  const ST* X;
  const ST* V;
  ST* X_out;
  ST* V_out;
  float* G;        // float32 dE/dX at the end point [n][P * 1024], or nullptr
  float* EX;
  float* EV;
  int64_t N, ntiles;
  int L;
  float eps, chalf;
};

template <typename ST>
struct SicEvalArgsT {
  const ST* X;
  float* G;         // float32 [Npad][1024] or nullptr
  float* E;
  float* EV;
  const ST* V;
  ST* V_gen;
  int64_t N, ntiles, first_pid;
  RngKey key;
};

using SicJumpArgs = SicJumpArgsT<__bf16>;
using SicLeapArgs = SicLeapArgsT<__bf16>;
using SicEvalArgs = SicEvalArgsT<__bf16>;
void sic_launch_jump(const SicJumpArgsT<__bf16>& a, const SicModel& mdl, hipStream_t st);
void sic_launch_eval(const SicEvalArgsT<__bf16>& a, const SicModel& mdl, hipStream_t st);
void sic_launch_leap(const SicLeapArgsT<__bf16>& a, const SicModel& mdl, hipStream_t st);
void sic_launch_jump(const SicJumpArgsT<float>& a, const SicModel& mdl, hipStream_t st);
void sic_launch_eval(const SicEvalArgsT<float>& a, const SicModel& mdl, hipStream_t st);
void sic_launch_leap(const SicLeapArgsT<float>& a, const SicModel& mdl, hipStream_t st);
// particles per 32-column tile: a tile holds whole particles (P columns each)
inline int sic_particles_per_tile(int P) { return 32 / P; }

}  // namespace mjhmc
