// Argument blocks of the ProductOfT MFMA kernels (dense_pot.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elementwise.hpp"  // Control, RngKey, philox

namespace mjhmc {

constexpr int kPotDim = 512;  // ndims == nbasis, zero padded to 512

// device-resident model, float32, padded to 512 x 512
struct PotModel {
  const float* W1;     // [d][j] = W[d][j] / nu_j              (first GEMM:  u = W1^T x + cb)
  const float* W2T;    // [j][d] = W[d][j] * (nu_j + 1) / nu_j  (second GEMM: dE/dx = W2T^T phi(u))
  const float* cb;     // [j]    = b_j / nu_j
  const float* alpha;  // [j]    = (nu_j + 1) / 2   (0 for padded experts)
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
  float* Hwork;        // [Npad] H of the inverse-L proposal for this attempt (cached or freshly integrated)
  int* cold_list;      // [Npad] compacted indices of the cold particles
  int* cold_count;
  float* EX_out;
  float* EV_out;
  float* Hflf_out;
  double* dwell;
  double* dwell_ring;
  uint8_t* trans;
  const float* noise;   // replay normals [N][512] or nullptr
  const double* rexp;   // replay unit exponentials [3][N] or nullptr
  Control* ctl;
  unsigned long long* stats;
  int64_t N, Npad, ntiles, first_pid;
  int D, L, iter;
  float eps, chalf, r_keep, r_mix;
  double p_r;
  RngKey key;
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

void pot_launch_jump(const PotJumpArgs& a, const PotModel& mdl, hipStream_t st);
void pot_launch_eval(const PotEvalArgs& a, const PotModel& mdl, hipStream_t st);

}  // namespace mjhmc
