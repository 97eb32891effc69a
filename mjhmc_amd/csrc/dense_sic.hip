// SparseImageCode (sparse-coding posterior over coefficients, mjhmc/misc/tf_distributions.py:204-272) on
// the bf16 matrix cores: bf16 state in HBM, bf16 MFMA operands, fp32 accumulation and fp32 integrator
// registers (BASELINE.json configs[4]).  img_size = 256; n_coeffs = 1024 or 512 (the two dictionaries the reference
// accepts, tf_distributions.py:219; template parameter NB = n_coeffs / 256; the text below is written for 1024);
// P = n_patches patches per particle (the reference's default is 9, :208; BASELINE configs[4] is one):
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
#include <cstdlib>

#include "dense_sic.hpp"

namespace mjhmc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

constexpr int kP = 32;
constexpr int kI = kSicImg;     // 256
// NB = 32-row blocks of X and V per wave = n_coeffs / 256 (4: the 1024-atom dictionary, 2: the 512-atom one)
#define kC (256 * NB)

__device__ __forceinline__ int acc_row(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

template <int NB>
struct CTile {
  f32x16 b[NB];  // this wave's 32 NB coefficient rows x 32 particles
};

// bf16 state rows [*][1024]: lane (c, h) reads its 16 groups of 4 consecutive coefficients (8 bytes each)
template <int NB>
__device__ __forceinline__ void ctile_load(const __bf16* base, int64_t p, int w, int h, CTile<NB>& t) {
  const __bf16* row = base + (size_t)p * kC + 32 * NB * w + 4 * h;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bf16x4 v = *reinterpret_cast<const bf16x4*>(row + 32 * b + 8 * g);
#pragma unroll
      for (int k = 0; k < 4; ++k) t.b[b][4 * g + k] = (float)v[k];
    }
}

template <int NB>
__device__ __forceinline__ void ctile_store(__bf16* base, int64_t p, int w, int h, const CTile<NB>& t) {
  __bf16* row = base + (size_t)p * kC + 32 * NB * w + 4 * h;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = (__bf16)t.b[b][4 * g + k];
      *reinterpret_cast<bf16x4*>(row + 32 * b + 8 * g) = v;
    }
}

constexpr int kRing = 8;    // dictionary fragments (1 KB each) a wave keeps in flight / staged
constexpr int kYLds = 12;   // patches whose pixels fit the LDS copy (more: read from global memory)

// ALL of the workgroup's LDS is this one object (a second __shared__ object beside an LDS-DMA target makes hipcc
// drain the DMA queue before LDS reads, cdna_hip_programming.md section 5)
template <int NB>
struct SicShared {
  f32x4 pubA[8][NB][2][64];   // 64 KB: a as B fragments, [wave][block][k-step half][lane]
  f32x4 pubR[8][2][64];       // 16 KB: scaled residual as B fragments
  f32x4 ring[8][kRing][64];   // 64 KB: per-wave ring of dictionary (A operand) fragments, filled by LDS-DMA
  float ys[kYLds * kI];       // 12 KB: the patches
  int ypatch[kP];             // byte offset of column c's patch inside ys (the column -> patch map does not depend on the tile)
  float red[2][8][kP];
  float colsum[kP];           // per-column energies, summed over a particle's columns when n_patches > 1
  int move[kP];
  unsigned tally[4];
};

// ---- the dictionary stream -----------------------------------------------------------------------------------------
// Both GEMMs of a leapfrog step read the whole dictionary (512 KB in each fragment order) from L2, 64 KB per wave
// and GEMM, one 1 KB fragment per MFMA.  Loading a fragment into registers right before its MFMA exposes a full L2
// round trip per MFMA (and every register spent on prefetch spills elsewhere: the kernel sits at the 256-VGPR budget).
// So the fragments go through LDS instead: `global_load_lds_dwordx4` (no VGPR destination) fills a per-wave ring of
// kRing slots, kRing fragments ahead of the MFMA that consumes them.  The fragment sequence of a wave is fixed --
// GEMM1 (64 fragments), GEMM2 (64), GEMM1, ... -- so the ring runs ahead ACROSS the GEMMs, the barriers between them
// and the tiles of the persistent loop: L2 latency is paid once per kernel.  Each wave reads only the slots it filled
// itself: ordering is the wave's own counted `s_waitcnt vmcnt(kRing - 1)`, no barrier.  The DMA is inline asm, so
// hipcc neither counts it nor drains it at `__syncthreads()`; its own waits (for loads it does count) can only
// over-wait, because the counter completes in issue order.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}
// lane l's 16 bytes at (sbase + voff) land at lds_dst + 16 l.  sbase is wave-uniform (scalar registers: its arithmetic
// is free), voff the lane's byte offset (ONE loop-invariant VGPR): per-lane 64-bit source pointers would cost two
// VGPRs per in-flight address, spill, and every spill reload is a counted load that drains the ring
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

constexpr unsigned kFrag1 = 2 * kI * 16;  // bytes between consecutive GEMM1 fragments (k-steps) of a lane
#define kFrag2 (2u * kC * 16u)            // bytes between consecutive GEMM2 k-steps; blocks of a k-step are 512 B apart

struct AStream {
  const char* a1;   // GEMM1 fragments: A1[k-step][h][image row][8 bf16]           (wave-uniform base)
  const char* a2;   // GEMM2 fragments: A2[k-step][h][coefficient row][8 bf16]
  unsigned v1, v2;  // this lane's byte offset inside a GEMM1 / GEMM2 fragment
  unsigned lds0;    // LDS byte address of this wave's ring
  int only1;        // the kernel runs GEMM1 only (an evaluation without gradient): the sequence is GEMM1, GEMM1, ...
};

// GEMM1 fragment (k-step) `pos` < kRing into ring slot pos: the head of the stream
__device__ __forceinline__ void astream_issue(const AStream& s, int pos) {
  glds16(s.a1 + (size_t)pos * kFrag1, s.v1, s.lds0 + (unsigned)pos * 1024u);
}

template <int NB>
__device__ __forceinline__ AStream astream_open(const SicModel& mdl, SicShared<NB>& sh, int w, int c, int h, bool only1) {
  AStream s;
  const size_t copy = (size_t)((blockIdx.x >> 3) % (unsigned)mdl.copies) * (size_t)(kI * kC * 2);  // blocks b, b + 8 share an XCD
  s.a1 = reinterpret_cast<const char*>(mdl.A1) + copy;
  s.a2 = reinterpret_cast<const char*>(mdl.A2) + copy;
  s.v1 = (unsigned)(h * kI + 32 * w + c) * 16u;
  s.v2 = (unsigned)(h * kC + 32 * NB * w + c) * 16u;
  s.lds0 = __builtin_amdgcn_readfirstlane(lds_addr(&sh.ring[w][0][0]));
  s.only1 = only1 ? 1 : 0;
#pragma unroll
  for (int j = 0; j < kRing; ++j) astream_issue(s, j);  // the first GEMM1 fragments
  return s;
}

// before the wave exits: no DMA may still be writing LDS that the next workgroup will own
__device__ __forceinline__ void astream_close() { wait_vm<0>(); }

// a GEMM1 whose GEMM2 is not run (a trajectory of zero steps): put the ring back on GEMM1's first fragments
__device__ __forceinline__ void astream_rewind(const AStream& s) {
  wait_vm<0>();
#pragma unroll
  for (int j = 0; j < kRing; ++j) astream_issue(s, j);
}

// the patches into LDS (once per kernel)
template <int NB>
__device__ __forceinline__ void stage_patches(const SicModel& mdl, SicShared<NB>& sh) {
  if (mdl.P <= kYLds)
    for (int i = threadIdx.x; i < mdl.P * kI; i += blockDim.x) sh.ys[i] = mdl.y[i];
  if (threadIdx.x < kP) {  // col_of: column c works on patch min(c, cpt - 1) % P
    const int cpt = (kP / mdl.P) * mdl.P;
    sh.ypatch[threadIdx.x] = ((threadIdx.x < cpt ? (int)threadIdx.x : cpt - 1) % mdl.P) * kI * (int)sizeof(float);
  }
  __syncthreads();
}

// this lane's index in its wave, produced by instructions the compiler must re-issue wherever it is asked for: addresses
// derived from it are RECOMPUTED inside the leapfrog loop (two vector instructions) instead of being kept live across it
// -- at the 256-register budget they were spilled, and every spill reload is a counted load that drains the ring
__device__ __forceinline__ unsigned lane_id_here() {
  unsigned l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

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
template <int NB, bool YG>
__device__ __forceinline__ void sic_residual(const SicModel& mdl, SicShared<NB>& sh, const AStream& as, int w, int c, int h,
                                             int lane, int patch, const CTile<NB>& x, f32x16& res) {
  constexpr int kSteps = 16 * NB;  // GEMM1 k-steps = fragments per wave
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    sh.pubA[w][b][0][lane] = frag_of(x.b[b], 0, 1.0f);
    sh.pubA[w][b][1][lane] = frag_of(x.b[b], 1, 1.0f);
  }
  __syncthreads();
  {  // res starts at -y[i]
    using lds_f32x4 = __attribute__((address_space(3))) const f32x4;
    using lds_int = __attribute__((address_space(3))) const int;
    using glb_f32x4 = __attribute__((address_space(1))) const f32x4;
    if constexpr (!YG) {
      // the patches are in LDS (n_patches <= kYLds): ds_read_b128 from an address rebuilt here from the lane index --
      // nothing counted in vmcnt, nothing kept in a register across the loop.  (A pointer selected at run time between
      // LDS and global memory is generic: its loads are flat_load, waited for with vmcnt(0), which drained the
      // dictionary ring at the head of every GEMM1.)
      const unsigned lid = lane_id_here();
      unsigned ya = lds_addr(sh.ys) + (unsigned)__builtin_amdgcn_readfirstlane(128 * w) + ((lid >> 5) << 4);
      if (mdl.P > 1) ya += (unsigned)*(lds_int*)(unsigned long)(lds_addr(sh.ypatch) + ((lid & 31u) << 2));
      lds_f32x4* yv = (lds_f32x4*)(unsigned long)ya;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = yv[2 * g];
#pragma unroll
        for (int k = 0; k < 4; ++k) res[4 * g + k] = -v[k];
      }
    } else {  // n_patches 13..32: global loads (counted: this instantiation drains the ring once per GEMM1)
      glb_f32x4* yv = (glb_f32x4*)(__attribute__((address_space(1))) const float*)(mdl.y + kI * patch + 32 * w + 4 * h);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = yv[2 * g];
#pragma unroll
        for (int k = 0; k < 4; ++k) res[4 * g + k] = -v[k];
      }
    }
  }
  // 16 NB k-steps; the fragment of k-step ks is in ring slot ks % kRing, and its slot is refilled with the fragment
  // kRing positions further down the stream as soon as it has been read.  The LDS reads of k-step ks + 1 are issued
  // before the MFMA of k-step ks (one register set ahead), so an MFMA never waits for the LDS round trip.
  const f32x4* ring = &sh.ring[w][0][lane];
  f32x4 fa, fb, na, nb;
  wait_vm<kRing - 1>();
  fa = ring[0];
  fb = sh.pubA[0][0][0][lane];
  auto chunk = [&](int ch, const char* refill, unsigned voff, unsigned step, unsigned step_hi, bool last) {
    // refill: wave-uniform address of the fragment that goes into slot 0; slot j gets refill + (j % NB) step + (j / NB) step_hi
#pragma unroll
    for (int j = 0; j < kRing; ++j) {
      const int ks = ch * kRing + j;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fa, fb are here; slot j has been read: it may be refilled
      __builtin_amdgcn_sched_barrier(0);
      glds16(refill + (j % NB) * step + (j / NB) * step_hi, voff, as.lds0 + (unsigned)j * 1024u);
      if (!(last && j == kRing - 1)) {
        wait_vm<kRing - 1>();  // fragment ks + 1 has landed
        const int kn = ks + 1;
        na = ring[((j + 1) & (kRing - 1)) * 64];
        nb = sh.pubA[kn / (2 * NB)][(kn >> 1) % NB][kn & 1][lane];
      }
      __builtin_amdgcn_sched_barrier(0);
      res = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), res,
                                                    0, 0, 0);
      fa = na;
      fb = nb;
    }
  };
#pragma unroll 1
  for (int ch = 0; ch < kSteps / kRing - 1; ++ch)
    chunk(ch, as.a1 + (size_t)(ch + 1) * kRing * kFrag1, as.v1, kFrag1, NB * kFrag1, false);
  if (as.only1) chunk(kSteps / kRing - 1, as.a1, as.v1, kFrag1, NB * kFrag1, true);  // the next GEMM1's first fragments
  else chunk(kSteps / kRing - 1, as.a2, as.v2, 512, kFrag2, true);  // GEMM2's first: (k-step 0, blocks 0..NB-1), (k-step 1, ...), ...
}

// acc[c][n] += sum_i B[i][c] * (scale * res[i][n]) + scale * prior'(x)   (GEMM2 into the caller's tile)
template <bool CAUCHY, int NB>
__device__ __forceinline__ void sic_kick(const SicModel& mdl, SicShared<NB>& sh, const AStream& as, int w, int c, int h,
                                         int lane, const f32x16& res, const CTile<NB>& x, float scale, CTile<NB>& acc) {
  constexpr int kPos = 16 * NB;         // GEMM2 stream positions per wave: 16 k-steps x NB blocks
  constexpr int kKs = kRing / NB;       // k-steps per chunk of kRing positions
  sh.pubR[w][0][lane] = frag_of(res, 0, scale * mdl.invP);  // d/da_p of the MEAN over patches
  sh.pubR[w][1][lane] = frag_of(res, 1, scale * mdl.invP);
  {  // the prior's force first (x is not read again until the drift): lambda * 2a / (1 + a^2), or lambda * sign(a)
    const float sl = scale * mdl.lambda;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float a = x.b[b][q];
        if (CAUCHY) acc.b[b][q] += (sl * 2.0f) * a * __builtin_amdgcn_rcpf(1.0f + a * a);  // v_rcp_f32: 1 ulp
        else acc.b[b][q] += sl * (a > 0.f ? 1.0f : (a < 0.f ? -1.0f : 0.0f));
      }
  }
  __syncthreads();
  const f32x4* ring = &sh.ring[w][0][lane];
  // stream positions 8 ch + j = (k-step kKs ch + j / NB, block j % NB); refilled kRing positions ahead: the same
  // (j / NB, block) of chunk ch + 1, or, from the last chunk, the next GEMM1's first fragments.  The dictionary fragment
  // of position + 1 is read from the ring before the MFMA of this position.
  f32x4 fa, na;
  wait_vm<kRing - 1>();
  fa = ring[0];
  auto chunk = [&](int ch, const char* refill, unsigned voff, unsigned step, unsigned step_hi, bool last) {
    f32x4 fb[kKs];
#pragma unroll
    for (int t = 0; t < kKs; ++t) fb[t] = sh.pubR[(kKs * ch + t) >> 1][(kKs * ch + t) & 1][lane];
#pragma unroll
    for (int j = 0; j < kRing; ++j) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fa (and fb) are here; slot j may be refilled
      __builtin_amdgcn_sched_barrier(0);
      glds16(refill + (j % NB) * step + (j / NB) * step_hi, voff, as.lds0 + (unsigned)j * 1024u);
      if (!(last && j == kRing - 1)) {
        wait_vm<kRing - 1>();
        na = ring[((j + 1) & (kRing - 1)) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
      acc.b[j % NB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa),
                                                              __builtin_bit_cast(bf16x8, fb[j / NB]), acc.b[j % NB], 0, 0, 0);
      fa = na;
    }
  };
#pragma unroll 1
  for (int ch = 0; ch < kPos / kRing - 1; ++ch) chunk(ch, as.a2 + (size_t)(kKs * (ch + 1)) * kFrag2, as.v2, 512, kFrag2, false);
  chunk(kPos / kRing - 1, as.a1, as.v1, kFrag1, NB * kFrag1, true);  // the next GEMM1's fragments 0..7: consecutive k-steps
}

__device__ __forceinline__ float half_swap_sum(float s) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(s), __float_as_int(s), false, false);
  return __int_as_float(sw[0]) + __int_as_float(sw[1]);
}

// sum of a per-column value over the columns of the caller's particle (n_patches of them, consecutive)
template <int NB>
__device__ __forceinline__ float group_total(SicShared<NB>& sh, int w, int c, int h, int P, int g0, float col_tot) {
  if (P == 1) return col_tot;
  if (w == 0 && h == 0) sh.colsum[c] = col_tot;
  __syncthreads();
  float t = 0.f;
  for (int j = 0; j < P; ++j) t += sh.colsum[g0 + j];
  __syncthreads();
  return t;
}

// E(x) per PARTICLE from the residuals at x:  mean_p 1/2 |res_p|^2 + lambda * prior(x)  (tf_distributions.py:257-270)
template <bool CAUCHY, int NB>
__device__ __forceinline__ float sic_energy(const SicModel& mdl, SicShared<NB>& sh, int w, int c, int h, const Col& col,
                                            const f32x16& res, const CTile<NB>& x) {
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += 0.5f * res[q] * res[q];
  float pr = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b)
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
template <int NB>
__device__ __forceinline__ float sic_kinetic(SicShared<NB>& sh, int w, int c, int h, int P, const Col& col, const CTile<NB>& v) {
  float s = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b)
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

template <int NB>
__device__ __forceinline__ void round_to_state(CTile<NB>& t) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      // opaque: otherwise hipcc shares these conversions with the ones that published the tile as an MFMA operand two
      // GEMMs earlier and keeps 64 single bf16 values alive (spilled) across both of them
      asm volatile("" : "+v"(t.b[b][q]));
      t.b[b][q] = (float)(__bf16)t.b[b][q];
    }
}

// L leapfrog steps with the half kicks between drifts merged (bf16 operands make the reference's
// separate roundings meaningless).  Returns E(x_new) of the particle; x, v updated in place.
template <bool CAUCHY, int NB, bool YG>
__device__ __forceinline__ float sic_trajectory(const SicModel& mdl, SicShared<NB>& sh, const AStream& as, int w, int c, int h,
                                                int lane, const Col& col, CTile<NB>& x, CTile<NB>& v, int L, float eps,
                                                float chalf) {
  // the step scale is wave-uniform: both values sit in scalar registers and the select is an integer s_cselect (a float
  // select is a per-lane v_cndmask whose operand was spilled and reloaded -- a counted load -- in front of every GEMM2)
  const int c_half = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, chalf));
  const int c_full = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 2.0f * chalf));
  f32x16 res;
  sic_residual<NB, YG>(mdl, sh, as, w, c, h, lane, col.patch, x, res);
  if (L > 0) sic_kick<CAUCHY, NB>(mdl, sh, as, w, c, h, lane, res, x, chalf, v);
  else astream_rewind(as);
  for (int s = 1; s <= L; ++s) {
    asm volatile("; MJHMC_LEAPFROG_STEP_BEGIN (tools/check_isa.sh)");
#pragma unroll
    for (int b = 0; b < NB; ++b) x.b[b] = x.b[b] + eps * v.b[b];
    sic_residual<NB, YG>(mdl, sh, as, w, c, h, lane, col.patch, x, res);
    // the step scale is wave-uniform: selected as an integer so that it stays in a scalar register (a float select is a
    // per-lane v_cndmask whose operand was spilled and reloaded -- a counted load -- in front of every GEMM2)
    sic_kick<CAUCHY, NB>(mdl, sh, as, w, c, h, lane, res, x, __builtin_bit_cast(float, s < L ? c_full : c_half), v);
    asm volatile("; MJHMC_LEAPFROG_STEP_END");
  }
  // the successor position is stored in bf16 (and GEMM1 already saw bf16(x)): evaluate the prior on
  // what will be stored, so EX is the energy of the stored state
  round_to_state(x);
  return sic_energy<CAUCHY, NB>(mdl, sh, w, c, h, col, res, x);
}

// v += mix * (standard normals of this lane's dims of particle `pid`; dims patch * 1024 + ...), group by group: one
// Box-Muller quadruple's temporaries at a time keeps this rare branch from dictating the kernel's register budget
template <int NB>
__device__ __forceinline__ void sic_add_normals(const RngKey& key, uint32_t pid, int patch, int w, int h, float mix,
                                                CTile<NB>& v) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll 1
    for (int g4 = 0; g4 < 4; ++g4) {
      const int d = kC * patch + 32 * NB * w + 32 * b + 8 * g4 + 4 * h;
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
template <bool CAUCHY, int NB, bool YG>
__global__ __launch_bounds__(512, 2) void sic_eval_kernel(const SicEvalArgs a, const SicModel mdl) {
  __shared__ SicShared<NB> sh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  stage_patches(mdl, sh);
  const AStream as = astream_open(mdl, sh, w, c, h, a.G == nullptr);
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const Col col = col_of(tile, c, mdl.P, a.N, Identity{});
    CTile<NB> x;
    ctile_load(a.X, col.q, w, h, x);
    f32x16 res;
    sic_residual<NB, YG>(mdl, sh, as, w, c, h, lane, col.patch, x, res);
    if (a.G) {
      CTile<NB> g;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) g.b[b][q] = 0.f;
      sic_kick<CAUCHY, NB>(mdl, sh, as, w, c, h, lane, res, x, 1.0f, g);
      if (col.alive) {
        float* row = a.G + (size_t)col.q * kC + 32 * NB * w + 4 * h;  // dE/dX is handed out in float32
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = g.b[b][4 * gq + k];
            *reinterpret_cast<f32x4*>(row + 32 * b + 8 * gq) = o;
          }
      }
    }
    const float ex = sic_energy<CAUCHY, NB>(mdl, sh, w, c, h, col, res, x);
    if (a.E && w == 0 && h == 0 && col.leader) a.E[col.part] = ex;
    if (a.EV) {
      CTile<NB> v;
      if (a.V_gen) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
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
  astream_close();
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

template <bool CAUCHY, int NB, bool YG>
__global__ __launch_bounds__(512, 2) void sic_flf_kernel(const SicJumpArgs a, const SicModel mdl) {
  __shared__ SicShared<NB> sh;
  if (a.ctl->failed) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  const int ncold = *a.cold_count;
  const int ppt = kP / mdl.P;
  if ((int64_t)blockIdx.x * ppt >= ncold) return;  // nothing for this workgroup
  stage_patches(mdl, sh);
  const AStream as = astream_open(mdl, sh, w, c, h, false);
  for (int64_t tile = blockIdx.x; tile * ppt < ncold; tile += gridDim.x) {
    const Col col = col_of(tile, c, mdl.P, (int64_t)ncold, FromList{a.cold_list});  // the last tile repeats an entry
    CTile<NB> x, v;
    ctile_load(a.X_in, col.q, w, h, x);
    ctile_load(a.V_in, col.q, w, h, v);
#pragma unroll
    for (int b = 0; b < NB; ++b) v.b[b] = -v.b[b];
    const float ex = sic_trajectory<CAUCHY, NB, YG>(mdl, sh, as, w, c, h, lane, col, x, v, a.L, a.eps, a.chalf);
    round_to_state(v);  // the same rounding the jump kernel applies to the forward proposal
    const float ev = sic_kinetic(sh, w, c, h, mdl.P, col, v);
    if (w == 0 && h == 0 && col.leader) a.Hwork[col.part] = ex + ev;
    __syncthreads();
  }
  astream_close();
}

// MODE = kModeMJHMC (markov_jump_hmc.py:355-415), kModeCT (:251-290) or kModeControl (:116-148, the comparison arm of
// the reference's sparse-coding experiments, search/control_sp_img/control_objective.py:10)
template <bool CAUCHY, bool REPLAY, int MODE, int NB, bool YG>
__global__ __launch_bounds__(512, 2) void sic_jump_kernel(const SicJumpArgs a, const SicModel mdl) {
  __shared__ SicShared<NB> sh;
  if (a.ctl->failed) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0;  // tallies (meaning per mode: fill_iter_stats in api.hip)
  bool any_bad = false;
  if (threadIdx.x < 4) sh.tally[threadIdx.x] = 0;
  stage_patches(mdl, sh);
  const AStream as = astream_open(mdl, sh, w, c, h, false);
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const Col col = col_of(tile, c, mdl.P, a.N, Identity{});
    const int64_t p = col.part;
    const float EX0 = a.EX_in[p], EV0 = a.EV_in[p];
    const float H0 = EX0 + EV0;
    const float Hflf = MODE == kModeMJHMC ? a.Hwork[p] : 0.f;
    CTile<NB> x, v;
    ctile_load(a.X_in, col.q, w, h, x);
    ctile_load(a.V_in, col.q, w, h, v);
    const float EXL = sic_trajectory<CAUCHY, NB, YG>(mdl, sh, as, w, c, h, lane, col, x, v, a.L, a.eps, a.chalf);
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
        for (int b = 0; b < NB; ++b) v.b[b] = -v.b[b];
      }
      if (k & 2) {
#pragma unroll
        for (int b = 0; b < NB; ++b) v.b[b] = -v.b[b];
      }
      refresh = (mv & 4) != 0;  // batch-wide (markov_jump_hmc.py:138-141)
    } else {
      if (k != 0) {  // F / R keep the position
        ctile_load(a.X_in, col.q, w, h, x);
        ctile_load(a.V_in, col.q, w, h, v);
      }
      if ((MODE == kModeCT && k == 0) || k == 1) {  // CT's FL move ends with a flip (:258,278); F flips
#pragma unroll
        for (int b = 0; b < NB; ++b) v.b[b] = -v.b[b];
      }
      refresh = (k == 2);
    }
    const bool tile_refreshes = __ballot(refresh) != 0ull;
    if (refresh) {  // HMCState.R (hmc_state.py:121-129)
#pragma unroll
      for (int b = 0; b < NB; ++b) v.b[b] = v.b[b] * a.r_keep;
      if constexpr (REPLAY) {
        const __bf16* zrow = a.noise + (size_t)col.q * kC + 32 * NB * w + 4 * h;
#pragma unroll
        for (int b = 0; b < NB; ++b)
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
  astream_close();
  if (n0) atomicAdd(&sh.tally[0], n0);
  if (n1) atomicAdd(&sh.tally[1], n1);
  if (n2) atomicAdd(&sh.tally[2], n2);
  if (n3) atomicAdd(&sh.tally[3], n3);
  __syncthreads();
  if (threadIdx.x < 4 && sh.tally[threadIdx.x]) atomicAdd(&a.stats[threadIdx.x], (unsigned long long)sh.tally[threadIdx.x]);
}

// HMCState.leapfrog / HMCState.L on caller-supplied states (hmc_state.py:86-100)
template <bool CAUCHY, int NB, bool YG>
__global__ __launch_bounds__(512, 2) void sic_leap_kernel(const SicLeapArgs a, const SicModel mdl) {
  __shared__ SicShared<NB> sh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  stage_patches(mdl, sh);
  const AStream as = astream_open(mdl, sh, w, c, h, false);
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const Col col = col_of(tile, c, mdl.P, a.N, Identity{});
    CTile<NB> x, v;
    ctile_load(a.X, col.q, w, h, x);
    ctile_load(a.V, col.q, w, h, v);
    const float ex = sic_trajectory<CAUCHY, NB, YG>(mdl, sh, as, w, c, h, lane, col, x, v, a.L, a.eps, a.chalf);
    round_to_state(v);
    const float ev = sic_kinetic(sh, w, c, h, mdl.P, col, v);
    if (a.G) {  // dE/dX of the stored end point
      f32x16 res;
      sic_residual<NB, YG>(mdl, sh, as, w, c, h, lane, col.patch, x, res);
      CTile<NB> g;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) g.b[b][q] = 0.f;
      sic_kick<CAUCHY, NB>(mdl, sh, as, w, c, h, lane, res, x, 1.0f, g);
      if (col.alive) {
        float* row = a.G + (size_t)col.q * kC + 32 * NB * w + 4 * h;
#pragma unroll
        for (int b = 0; b < NB; ++b)
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
  astream_close();
}

static int sic_cus() {
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return std::max(1, cus);
}

template <bool CAUCHY, int MODE, int NB, bool YG>
static void sic_launch_mode(const SicJumpArgs& a, const SicModel& mdl, unsigned grid, hipStream_t st) {
  const bool replay = MODE == kModeControl ? (a.runif && a.noise) : (a.rexp && a.noise);
  if (replay) hipLaunchKernelGGL((sic_jump_kernel<CAUCHY, true, MODE, NB, YG>), dim3(grid), dim3(512), 0, st, a, mdl);
  else hipLaunchKernelGGL((sic_jump_kernel<CAUCHY, false, MODE, NB, YG>), dim3(grid), dim3(512), 0, st, a, mdl);
}

template <bool CAUCHY, int NB, bool YG>
static void sic_launch_jump_t(const SicJumpArgs& a, const SicModel& mdl, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, sic_cus());
  if (a.mode == kModeMJHMC) {  // only MJHMC has the inverse-L proposal and its cache
    (void)hipMemsetAsync(a.cold_count, 0, sizeof(int), st);
    hipLaunchKernelGGL(sic_cold_list_kernel, dim3((unsigned)((a.Npad + 255) / 256)), dim3(256), 0, st, a.Hflf_in, a.Hwork,
                       a.N, a.Npad, a.cold_list, a.cold_count, (const Control*)a.ctl, a.stats);
    hipLaunchKernelGGL((sic_flf_kernel<CAUCHY, NB, YG>), dim3(grid), dim3(512), 0, st, a, mdl);
    sic_launch_mode<CAUCHY, kModeMJHMC, NB, YG>(a, mdl, grid, st);
  } else if (a.mode == kModeCT) {
    sic_launch_mode<CAUCHY, kModeCT, NB, YG>(a, mdl, grid, st);
  } else {
    sic_launch_mode<CAUCHY, kModeControl, NB, YG>(a, mdl, grid, st);
  }
}

// the prior (Cauchy / Laplace), the dictionary width (1024 / 512 atoms) and where the patches are read from (LDS for
// n_patches <= kYLds, global memory otherwise) select the instantiation
#define SIC_DISPATCH(CALL)                                                 \
  do {                                                                     \
    const bool yg = mdl.P > kYLds;                                         \
    if (mdl.nc == 1024) {                                                  \
      if (mdl.cauchy) { if (yg) CALL(true, 4, true); else CALL(true, 4, false); }     \
      else { if (yg) CALL(false, 4, true); else CALL(false, 4, false); }               \
    } else {                                                               \
      if (mdl.cauchy) { if (yg) CALL(true, 2, true); else CALL(true, 2, false); }     \
      else { if (yg) CALL(false, 2, true); else CALL(false, 2, false); }               \
    }                                                                      \
  } while (0)

void sic_launch_jump(const SicJumpArgs& a, const SicModel& mdl, hipStream_t st) {
#define SIC_JUMP(C, NBV, YGV) sic_launch_jump_t<C, NBV, YGV>(a, mdl, st)
  SIC_DISPATCH(SIC_JUMP);
}

void sic_launch_eval(const SicEvalArgs& a, const SicModel& mdl, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, sic_cus());
#define SIC_EVAL(C, NBV, YGV) hipLaunchKernelGGL((sic_eval_kernel<C, NBV, YGV>), dim3(grid), dim3(512), 0, st, a, mdl)
  SIC_DISPATCH(SIC_EVAL);
}

void sic_launch_leap(const SicLeapArgs& a, const SicModel& mdl, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, sic_cus());
#define SIC_LEAP(C, NBV, YGV) hipLaunchKernelGGL((sic_leap_kernel<C, NBV, YGV>), dim3(grid), dim3(512), 0, st, a, mdl)
  SIC_DISPATCH(SIC_LEAP);
}

}  // namespace mjhmc
