// Argument blocks of the SparseImageCode bf16-MFMA kernels (dense_sic.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dense_pot.hpp"  // dense_decide, normal_pair_f32, Control, RngKey

namespace mjhmc {

constexpr int kSicCoeffs = 1024;  // n_coeffs (state dims, one patch per particle)
constexpr int kSicImg = 256;      // img_size

struct SicModel {
  const void* A1;   // bf16 [64 k-steps][2 halves][256 image rows][8]: B[i][c] in GEMM1's fragment k-order
  const void* A2;   // bf16 [16 k-steps][2 halves][1024 coeffs][8]:    B[i][c] in GEMM2's fragment k-order
  const float* y;   // [256] the patch
  float lambda;
  int cauchy;
};

struct SicJumpArgs {
  const __bf16* X_in;
  const __bf16* V_in;
  __bf16* X_out;
  __bf16* V_out;
  const float* EX_in;
  const float* EV_in;
  const float* Hflf_in;
  float* Hwork;
  int* cold_list;
  int* cold_count;
  float* EX_out;
  float* EV_out;
  float* Hflf_out;
  double* dwell;
  double* dwell_ring;
  uint8_t* trans;
  const __bf16* noise;
  const double* rexp;
  Control* ctl;
  unsigned long long* stats;
  int64_t N, Npad, ntiles, first_pid;
  int L, iter;
  float eps, chalf, r_keep, r_mix;
  double p_r;
  RngKey key;
};

struct SicEvalArgs {
  const __bf16* X;
  float* G;         // float32 [Npad][1024] or nullptr
  float* E;
  float* EV;
  const __bf16* V;
  __bf16* V_gen;
  int64_t N, ntiles, first_pid;
  RngKey key;
};

void sic_launch_jump(const SicJumpArgs& a, const SicModel& mdl, hipStream_t st);
void sic_launch_eval(const SicEvalArgs& a, const SicModel& mdl, hipStream_t st);

}  // namespace mjhmc
