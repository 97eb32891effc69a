// Building blocks of the ProductOfT matrix-core kernels (dense_pot.hip: float32 state; dense_pot64.hip: the reference's
// arithmetic, float64 state around the float32 force): the accumulator-layout tile, its LDS image, the software-pipelined
// GEMM over the L2-resident pre-scaled matrices, the gradient evaluation built from two of them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dense_pot.hpp"
#include "timing_variants.hpp"   // POT_STAMP: cycle stamps of the timing build (tools/pot_stamps.sh); nothing otherwise

namespace mjhmc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kP = 32;  // particles per tile

// The padded dimension DIM = 128 * NB (NB = 1, 2, 4 accumulator blocks per wave): ndims == nbasis <= 128, 256, 512.
// Wave w owns rows [32 NB w, 32 NB (w+1)); (block r < NB, reg q, lane half h) <-> row 32 NB w + NB * acc_row(q, h) + r,
// so a lane's NB blocks hold NB CONSECUTIVE rows per register index: one 4*NB-byte load/store each.
__device__ __forceinline__ int acc_row(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

template <int NB>
struct Tile {
  f32x16 b[NB];
};

template <int NB>
struct VecN;
template <>
struct VecN<1> {
  using type = float;
};
template <>
struct VecN<2> {
  using type = __attribute__((ext_vector_type(2))) float;
};
template <>
struct VecN<4> {
  using type = f32x4;
};

template <int NB>
__device__ __forceinline__ float vget(const typename VecN<NB>::type& v, int r) {
  if constexpr (NB == 1) return v;
  else return v[r];
}
template <int NB>
__device__ __forceinline__ void vset(typename VecN<NB>::type& v, int r, float x) {
  if constexpr (NB == 1) v = x;
  else v[r] = x;
}

// rows of particle `p` (particle-major [*, DIM] matrix): this lane's 16 groups of NB consecutive dims
template <int NB>
__device__ __forceinline__ void tile_load(const float* base, int64_t p, int w, int h, Tile<NB>& t) {
  using V = typename VecN<NB>::type;
  const float* row = base + (size_t)p * (128 * NB) + 32 * NB * w;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const V v = *reinterpret_cast<const V*>(row + NB * acc_row(q, h));
#pragma unroll
    for (int r = 0; r < NB; ++r) t.b[r][q] = vget<NB>(v, r);
  }
}

template <int NB>
__device__ __forceinline__ void tile_store(float* base, int64_t p, int w, int h, const Tile<NB>& t) {
  using V = typename VecN<NB>::type;
  float* row = base + (size_t)p * (128 * NB) + 32 * NB * w;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    V v;
#pragma unroll
    for (int r = 0; r < NB; ++r) vset<NB>(v, r, t.b[r][q]);
    *reinterpret_cast<V*>(row + NB * acc_row(q, h)) = v;
  }
}

// LDS image of one wave's tile: [r][q/4][lane] x float4 (lane-linear 16 B: conflict-free b128)
template <int NB>
struct PubWave {
  f32x4 v[NB][4][64];
};

template <int NB>
__device__ __forceinline__ void publish(PubWave<NB>& dst, int lane, const Tile<NB>& t) {
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      f32x4 v;
      v[0] = t.b[r][4 * q4 + 0];
      v[1] = t.b[r][4 * q4 + 1];
      v[2] = t.b[r][4 * q4 + 2];
      v[3] = t.b[r][4 * q4 + 3];
      dst.v[r][q4][lane] = v;
    }
}

// acc[r][i][c] += sum over all DIM k-rows of  M[k][32 NB w + NB i + r] * B[k][c]
//   M   : row-major [DIM][DIM] matrix in global memory (pre-scaled W or W^T)
//   pub : the four waves' published B tiles (k-rows in accumulator layout)
// The k-range is walked in 4*NB chunks of 16 k-pairs (one published block (ws, r) each).  With one wave
// per SIMD nothing else hides the L2 latency of the A rows, so they are software-pipelined by hand:
// chunk n+1's sixteen A loads are issued (and fenced against sinking) before chunk n's 16*NB MFMAs start.
template <int NB>
__device__ __forceinline__ void a_chunk_load(const float* mlane, int chunk, typename VecN<NB>::type (&dst)[16]) {
  using V = typename VecN<NB>::type;
  constexpr int DIM = 128 * NB;
  const float* base = mlane + (size_t)(32 * NB * (chunk / NB) + (chunk % NB)) * DIM;
#pragma unroll
  for (int q = 0; q < 16; ++q) dst[q] = *reinterpret_cast<const V*>(base + (size_t)(NB * ((q & 3) + 8 * (q >> 2))) * DIM);
}

// The same through a buffer resource: address = scalar base + the lane's constant 32-bit offset (VGPR) + the row's scalar
// offset.  No vector instruction per load -- with 64-bit global addresses every load of the block in front of a chunk
// needs a v_add_co / v_addc pair and a wait state, and the block is EXPOSED: the wave issues no MFMA meanwhile (a GEMM
// took 77-78 000 cycles for 65 536 of MFMAs, 69 000 without the A loads; tools/pot_stamps.sh).
template <int NB>
__device__ __forceinline__ typename VecN<NB>::type buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  using V = typename VecN<NB>::type;
  if constexpr (NB == 4) return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  else if constexpr (NB == 2) return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
  else return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <int NB>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t a_rsrc(const float* M) {
  constexpr int DIM = 128 * NB;
  return __builtin_amdgcn_make_buffer_rsrc((void*)M, 0, DIM * DIM * 4, 0x00020000);
}
template <int NB>
__device__ __forceinline__ void a_chunk_load_buf(__amdgpu_buffer_rsrc_t r, unsigned lane_bytes, int chunk,
                                                 typename VecN<NB>::type (&dst)[16]) {
  constexpr int DIM = 128 * NB;
  const unsigned cbase = (unsigned)(32 * NB * (chunk / NB) + (chunk % NB)) * (unsigned)(DIM * 4);   // wave-uniform
#pragma unroll
  for (int q = 0; q < 16; ++q) dst[q] = buf_load<NB>(r, lane_bytes, cbase + (unsigned)(NB * ((q & 3) + 8 * (q >> 2)) * DIM * 4));
}

// B operands (published tiles) are read one group of four k-pairs AHEAD of the MFMAs that use them: with one wave per
// SIMD an LDS read issued right before its MFMA stalls the matrix pipe for the whole LDS round trip.  The four waves'
// images are contiguous, [chunk][q4][lane] x float4: group t = 4 chunk + q4.
// (Measured and dropped: issuing the next chunk's sixteen A loads one per MFMA group, from a scalar base, instead of
// in front of the chunk -- 1.5 % slower.)
template <int NB>
__device__ __forceinline__ void chunk_mfma(const f32x4* bbase, int chunk, int lane, const typename VecN<NB>::type (&a)[16],
                                           f32x4& bnext, Tile<NB>& acc) {
  constexpr int NG = 16 * NB;  // groups per GEMM
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const f32x4 b4 = bnext;
    const int t = 4 * chunk + q4 + 1;
    bnext = bbase[(size_t)(t < NG ? t : NG - 1) * 64 + lane];  // (the last one is a harmless re-read)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
#pragma unroll
      for (int r = 0; r < NB; ++r)
        acc.b[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(vget<NB>(a[4 * q4 + qq], r), b4[qq], acc.b[r], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The GEMM's form of it: with the OTHER register set's next rows requested one per group of NB MFMAs (buffer loads: a
// scalar add and the load, nothing for the vector pipe).  As a block of sixteen in front of the chunk the requests were
// exposed -- the wave issues no MFMA meanwhile -- and with 64-bit global addresses (a v_add_co / v_addc pair and a wait state
// per load) they cost whether blocked or spread (round 2: spread was the slower form).  A GEMM of 65 536 cycles of MFMAs:
// 77-78 000 cycles with global loads in a block, 72 500 with buffer loads in a block, 68 000 spread (tools/pot_stamps.sh).
template <int NB>
__device__ __forceinline__ void chunk_mfma_ld(const f32x4* bbase, int chunk, int lane, const typename VecN<NB>::type (&a)[16],
                                              f32x4& bnext, Tile<NB>& acc, __amdgpu_buffer_rsrc_t rs, unsigned lane_bytes,
                                              unsigned cbase, typename VecN<NB>::type (&aload)[16]) {
  constexpr int NG = 16 * NB, DIM = 128 * NB;
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const f32x4 b4 = bnext;
    const int t = 4 * chunk + q4 + 1;
    bnext = bbase[(size_t)(t < NG ? t : NG - 1) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int q = 4 * q4 + qq;
#pragma unroll
      for (int r = 0; r < NB; ++r)
        acc.b[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(vget<NB>(a[q], r), b4[qq], acc.b[r], 0, 0, 0);
      aload[q] = buf_load<NB>(rs, lane_bytes, cbase + (unsigned)(NB * ((q & 3) + 8 * (q >> 2)) * DIM * 4));
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// a0 enters holding chunk 0 of M (loaded before the previous epilogue, see AReg) and leaves holding chunk 0 of Mnext:
// the next GEMM's first A rows cross the epilogue and the barrier
template <int NB>
__device__ __forceinline__ void gemm_dim(const float* __restrict__ M, const float* __restrict__ Mnext, const PubWave<NB>* pub,
                                         int w, int c, int h, int lane, typename VecN<NB>::type (&a0)[16], Tile<NB>& acc) {
  constexpr int DIM = 128 * NB, NCH = 4 * NB;
  const size_t lane_off = (size_t)(4 * NB * h) * DIM + 32 * NB * w + NB * c;
  const unsigned lane_bytes = (unsigned)(lane_off * sizeof(float));
  const __amdgpu_buffer_rsrc_t rM = a_rsrc<NB>(M), rN = a_rsrc<NB>(Mnext);
  const f32x4* bbase = &pub[0].v[0][0][0];
  typename VecN<NB>::type a1[16];
  f32x4 bnext = bbase[lane];
#pragma unroll 1
  for (int chunk = 0; chunk < NCH; chunk += 2) {
    // chunk's MFMAs read a0 while chunk + 1's rows are requested into a1, one per group of NB MFMAs; chunk + 1's read a1
    // while chunk + 2 (or chunk 0 of the next GEMM's matrix: it crosses the epilogue and the barrier) lands in a0.
    // One code path (selects, no branch: two paths made hipcc copy the 64 accumulator registers every chunk).
    const bool more = chunk + 2 < NCH;
    const int c1 = chunk + 1, c2 = more ? chunk + 2 : 0;
    chunk_mfma_ld<NB>(bbase, chunk, lane, a0, bnext, acc, rM, lane_bytes,
                      (unsigned)(32 * NB * (c1 / NB) + (c1 % NB)) * (unsigned)(DIM * 4), a1);
    chunk_mfma_ld<NB>(bbase, chunk + 1, lane, a1, bnext, acc, more ? rM : rN, lane_bytes,
                      (unsigned)(32 * NB * (c2 / NB) + (c2 % NB)) * (unsigned)(DIM * 4), a0);
  }
}

// ndims <= 128 (NB = 1): a wave's rows of BOTH pre-scaled matrices are 2 x 64 floats per lane -- they stay in
// registers for the whole kernel (the workgroup's 4 x 64 lanes x 128 VGPRs hold W1 and W2T completely), so a
// leapfrog step issues no global load at all.  This is the size of the reference's own ProductOfT experiments
// (36 dims, <= 1000 particles): a handful of tiles, one per CU, where nothing else could hide the A-row latency.
template <int NB>
struct AReg {  // NB > 1: chunk 0 of the matrix the NEXT GEMM reads (the GEMMs alternate W1, W2T, W1, ...), loaded ahead
  typename VecN<NB>::type a0[16];
};
template <>
struct AReg<1> {
  float w1[4][16];
  float w2[4][16];
};

template <int NB>
__device__ __forceinline__ void areg_load(const PotModel& mdl, int w, int c, int h, AReg<NB>& ar) {
  if constexpr (NB == 1) {
    const float* m1 = mdl.W1 + (size_t)(4 * h) * 128 + 32 * w + c;
    const float* m2 = mdl.W2T + (size_t)(4 * h) * 128 + 32 * w + c;
#pragma unroll
    for (int chunk = 0; chunk < 4; ++chunk) {
      a_chunk_load<1>(m1, chunk, ar.w1[chunk]);
      a_chunk_load<1>(m2, chunk, ar.w2[chunk]);
    }
  } else {
    a_chunk_load<NB>(mdl.W1 + (size_t)(4 * NB * h) * (128 * NB) + 32 * NB * w + NB * c, 0, ar.a0);
  }
}

template <int NB, bool SECOND>
__device__ __forceinline__ void gemm_any(const PotModel& mdl, AReg<NB>& ar, const PubWave<NB>* pub, int w, int c,
                                         int h, int lane, Tile<NB>& acc) {
  if constexpr (NB == 1) {
    // k-rows at or beyond ndims (== nbasis) are padding: X, phi(U) and the matrix rows are exactly zero there, so the
    // row groups of eight that lie entirely in the padding are skipped (36 dims: 20 of 64 MFMAs per GEMM remain)
    const int kdim = mdl.ndims;
#pragma unroll
    for (int chunk = 0; chunk < 4; ++chunk) {
      const float(&a)[16] = SECOND ? ar.w2[chunk] : ar.w1[chunk];
      const f32x4(*blk)[64] = pub[chunk].v[0];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        if (32 * chunk + 8 * q4 < kdim) {
          const f32x4 b4 = blk[q4][lane];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq)
            acc.b[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q4 + qq], b4[qq], acc.b[0], 0, 0, 0);
        }
      }
    }
  } else {
    gemm_dim<NB>(SECOND ? mdl.W2T : mdl.W1, SECOND ? mdl.W1 : mdl.W2T, pub, w, c, h, lane, ar.a0, acc);
  }
}

// per-row constant vector in accumulator layout
template <int NB>
__device__ __forceinline__ void rowvec_load(const float* vec, int w, int h, Tile<NB>& t) {
  using V = typename VecN<NB>::type;
  const float* base = vec + 32 * NB * w;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const V v = *reinterpret_cast<const V*>(base + NB * acc_row(q, h));
#pragma unroll
    for (int r = 0; r < NB; ++r) t.b[r][q] = vget<NB>(v, r);
  }
}

__device__ __forceinline__ float half_sum(float s) {  // + the other lane half
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(s), __float_as_int(s), false, false);
  return __int_as_float(sw[0]) + __int_as_float(sw[1]);
}

template <int NB>
struct Shared {
  PubWave<NB> pub[2][4];  // B operands, [0] = X tile, [1] = H tile (2 x 64 KB at NB = 4)
  float red[2][4][kP];    // per-wave partial sums (energy, kinetic)
  int move[kP];           // transition chosen per particle
  float cb[128 * NB];     // b_j / nu_j: the value every u accumulator starts from (stage_bias)
  float zn[128 * NB];     // the standard normals of ONE refreshing column (column_normals)
};

// What the kernels that only FINISH a move need of the above (pot_fix_kernel: no GEMM operands, no bias)
template <int NB>
struct FinishShared {
  float red[2][4][kP];
  int move[kP];
  float zn[128 * NB];
};

// HMCState.R (hmc_state.py:121-129) needs ndims standard normals for a column whose momentum is redrawn.  Drawn by the
// column's own lanes that is 32 float64 Box-Muller pairs per lane, one after the other, with the rest of the workgroup
// waiting (~20 us per tile with a refreshing column: a third of an iteration of the reference's own 36 x 1000 batches,
// 1.5 % of C3).  Instead ALL 256 threads draw one pair each for the column -- the same pairs (2 k, 2 k + 1) of the same
// counters, so the same values -- into LDS; the column's lanes read their elements after the barrier the caller places.
template <int NB, typename F>
__device__ __forceinline__ void column_normals(const RngKey& key, uint32_t pid, int D, F* zn) {
  const int pair = threadIdx.x;
  if (pair < 64 * NB) {
    double z0, z1;
    normal_pair(key, pid, (uint32_t)pair, z0, z1);
    zn[2 * pair] = 2 * pair < D ? (F)z0 : (F)0;
    zn[2 * pair + 1] = 2 * pair + 1 < D ? (F)z1 : (F)0;
  }
}

// b / nu into LDS, once per kernel.  Every gradient starts by initialising its 16 NB accumulator registers from it; as
// global loads these stood right behind the barrier in front of the first MFMA (and, in the float64-state kernel,
// behind the epilogue's stores in the in-order memory queue).  Call before the first barrier of the kernel.
template <int NB>
__device__ __forceinline__ void stage_bias(const PotModel& mdl, Shared<NB>& sh) {
  for (int i = threadIdx.x; i < 128 * NB; i += 256) sh.cb[i] = mdl.cb[i];
}


// gradient of the energy at the X held in `x`; optionally the energy itself.
// On return g holds dE/dX in the same layout as x.  Two barriers (X and H live in separate buffers:
// a wave can only reach the next publish of a buffer after every wave has passed the barrier that
// follows its last read of it).
// pot_gradient_published: the same from the barrier on -- the caller has written its part of the X image
// (sh.pub[0][w], publish()'s layout) already.
template <int NB>
__device__ __forceinline__ void pot_gradient_published(const PotModel& mdl, AReg<NB>& ar, Shared<NB>& sh, int w, int c, int h,
                                                       int lane, Tile<NB>& g, bool want_energy, float* energy_out,
                                                       int stamp_slot = 0) {
  __syncthreads();
  POT_STAMP(1);
  Tile<NB> u;
  rowvec_load<NB>(sh.cb, w, h, u);                     // u starts at b_j / nu_j (LDS copy, stage_bias)
  gemm_any<NB, false>(mdl, ar, sh.pub[0], w, c, h, lane, u);   // + sum_d W[d][j]/nu_j * x_d
  POT_STAMP(2);
  if (want_energy) {                                   // E = sum_j alpha_j log(1 + u_j^2)  (distributions.py:430-432)
    using V = typename VecN<NB>::type;
    const float* al = mdl.alpha + 32 * NB * w;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const V a4 = *reinterpret_cast<const V*>(al + NB * acc_row(q, h));
#pragma unroll
      for (int r = 0; r < NB; ++r) s += vget<NB>(a4, r) * logf(1.0f + u.b[r][q] * u.b[r][q]);
    }
    const float part = half_sum(s);
    if (h == 0) sh.red[0][w][c] = part;
  }
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float uu = u.b[r][q];
      u.b[r][q] = uu * __builtin_amdgcn_rcpf(1.0f + uu * uu);   // phi(u) (v_rcp_f32: 1 ulp); the factor (nu+1)/nu lives in W2T
    }
  publish<NB>(sh.pub[1][w], lane, u);
  POT_STAMP(3);
  __syncthreads();
  POT_STAMP(4);
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int q = 0; q < 16; ++q) g.b[r][q] = 0.f;
  gemm_any<NB, true>(mdl, ar, sh.pub[1], w, c, h, lane, g);
  POT_STAMP(5);
  if (want_energy && energy_out) {
    *energy_out = sh.red[0][0][c] + sh.red[0][1][c] + sh.red[0][2][c] + sh.red[0][3][c];
  }
}

template <int NB>
__device__ __forceinline__ void pot_gradient(const PotModel& mdl, AReg<NB>& ar, Shared<NB>& sh, int w, int c, int h,
                                             int lane, const Tile<NB>& x, Tile<NB>& g, bool want_energy,
                                             float* energy_out) {
  [[maybe_unused]] const int stamp_slot = 0;
  POT_STAMP(0);
  publish<NB>(sh.pub[0][w], lane, x);
  pot_gradient_published<NB>(mdl, ar, sh, w, c, h, lane, g, want_energy, energy_out);
}

// kinetic energy sum(v^2)/2 per particle (all lanes of column c get it).  One barrier pair.
template <int NB, class SH>
__device__ __forceinline__ float pot_kinetic(SH& sh, int w, int c, int h, const Tile<NB>& v) {
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < NB; ++r)
#pragma unroll
    for (int q = 0; q < 16; ++q) s += v.b[r][q] * v.b[r][q];
  const float part = half_sum(s);
  if (h == 0) sh.red[1][w][c] = part;
  __syncthreads();
  const float tot = (sh.red[1][0][c] + sh.red[1][1][c] + sh.red[1][2][c] + sh.red[1][3][c]) / 2.0f;
  __syncthreads();
  return tot;
}

// L leapfrog steps (hmc_state.py:86-100); g enters as dE/dX at x, leaves as dE/dX at the new x.
// Returns E(x_new) through *ex (the last gradient evaluation already has u(x_new)).
template <int NB>
__device__ __forceinline__ void pot_trajectory(const PotModel& mdl, AReg<NB>& ar, Shared<NB>& sh, int w, int c, int h,
                                               int lane, Tile<NB>& x, Tile<NB>& v, Tile<NB>& g, int L, float eps,
                                               float chalf, float* ex) {
  // the closing half kick of a step and the opening one of the next use the same gradient: one kick of twice the size
  // (float32 roundings differ from the reference's float64-state sequence either way; mjhmc/fast/hmc.py:6-98 merges too)
  if (L > 0) {
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) v.b[r][q] = v.b[r][q] + chalf * g.b[r][q];
  }
  for (int s = 0; s < L; ++s) {
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) x.b[r][q] = x.b[r][q] + eps * v.b[r][q];
    pot_gradient<NB>(mdl, ar, sh, w, c, h, lane, x, g, s == L - 1, ex);
    const float ck = s == L - 1 ? chalf : 2.0f * chalf;
#pragma unroll
    for (int r = 0; r < NB; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) v.b[r][q] = v.b[r][q] + ck * g.b[r][q];
  }
}

// standard normals for this lane's rows (Box-Muller pairs (2k, 2k+1) of the counter RNG), zero beyond D
template <int NB>
__device__ __forceinline__ void pot_normals(const RngKey& key, uint32_t pid, int w, int h, int D, Tile<NB>& z) {
#pragma unroll 1
  for (int q = 0; q < 16; ++q) {
    const int d = 32 * NB * w + NB * acc_row(q, h);
    double zz[4] = {0, 0, 0, 0};
    normal_pair(key, pid, (uint32_t)(d >> 1), zz[0], zz[1]);
    if constexpr (NB == 4) normal_pair(key, pid, (uint32_t)((d >> 1) + 1), zz[2], zz[3]);
    if constexpr (NB == 1) {
      z.b[0][q] = d < D ? (float)((d & 1) ? zz[1] : zz[0]) : 0.f;
    } else {
#pragma unroll
      for (int r = 0; r < NB; ++r) z.b[r][q] = d + r < D ? (float)zz[r] : 0.f;
    }
  }
}

}  // namespace mjhmc
