// ProductOfT in the REFERENCE'S arithmetic on the matrix cores: float64 HMCState arrays (hmc_state.py:29-38) integrated
// around a float32 force (distributions.py:398-415: float32 shared variables, allow_input_downcast=True).
//
//   V += (-eps/2) * dEdX ;  X += eps * V ;  dEdX = float64( force32( float32(X) ) ) ;  V += (-eps/2) * dEdX     (hmc_state.py:86-91)
//
// The tile kernel of dense_pot.hip keeps X, V as float32 accumulator tiles in registers.  Here they live in HBM / MALL
// as float64 rows and never as register tiles: the wave's 128 registers of x / v tiles become the STAGING registers of
// a streamed epilogue.  After the second GEMM of a gradient the wave walks its accumulator (dE/dX, float32) in four
// chunks of 16 elements per lane: load the float64 V and X elements of the chunk (the next chunk's loads are in flight
// while this one is worked on), V += c g (twice between two drifts: the closing half kick of a step and the opening one
// of the next are two roundings, as in the reference), X += eps V in float64, store both, and write float32(X) straight
// into the LDS B-fragment image the first GEMM of the next gradient reads.  dE/dX stays float32 (the reference's force
// output, widened exactly where the reference assigns it into its float64 dEdX array, hmc_state.py:52-53).
//
// Working copy: between the passes of a trajectory the wave's X and V elements live in a per-workgroup scratch (2 x 32
// rows = 256 KB per workgroup, 64 MB per launch: MALL-resident) in LANE-LINEAR 16-byte pieces, so the per-step passes are
// fully coalesced (see Work<NB>); the particle rows are read by the first pass and written by the last drift (X) and
// the closing kick (V) only.  A particle that takes the L move is finished when the trajectory is; one that does not
// gets its pre-move rows copied over at the end.
// Traffic: 32 particles x 512 dims x 8 B x (X, V) x (read + write) = 512 KB per tile and leapfrog step = 3.5 B/clk/CU
// beside ~146 000 cycles of gradient; at the CU's 64 B/clk vector-memory port that is 8 192 cycles per step.
//
// Same operations in the same order as the multi-pass path this replaces on the sampling path (host_energy.hip:
// hk_pot_kick_drift / pot_eval_kernel) and the same GEMM code: results are bit-identical to it
// (tests/test_gpu_dense_parity.py::test_pot64_fused_equals_multipass).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "dense_pot.hpp"
#include "dense_pot_tile.hpp"

namespace mjhmc {

template <int NB>
struct DVecN;
template <>
struct DVecN<1> {
  using type = double;
};
template <>
struct DVecN<2> {
  using type = __attribute__((ext_vector_type(2))) double;
};
template <>
struct DVecN<4> {
  using type = __attribute__((ext_vector_type(4))) double;
};
template <int NB>
__device__ __forceinline__ double dget(const typename DVecN<NB>::type& v, int r) {
  if constexpr (NB == 1) return v;
  else return v[r];
}
template <int NB>
__device__ __forceinline__ void dset(typename DVecN<NB>::type& v, int r, double x) {
  if constexpr (NB == 1) v = x;
  else v[r] = x;
}

// A lane's elements of particle row `row` ([*][DIM] float64): register index q <-> NB consecutive doubles at
// row + 32 NB w + NB acc_row(q, h).  The (w, h) part is folded into the lane's base pointer, the q part is an
// immediate offset of the load.
template <int NB>
__device__ __forceinline__ int q_off(int q) { return NB * ((q & 3) + 8 * (q >> 2)); }
template <int NB>
__device__ __forceinline__ const double* lane_row(const double* base, int64_t p, int w, int h) {
  return base + (size_t)p * (128 * NB) + 32 * NB * w + 4 * NB * h;
}
template <int NB>
__device__ __forceinline__ double* lane_row(double* base, int64_t p, int w, int h) {
  return base + (size_t)p * (128 * NB) + 32 * NB * w + 4 * NB * h;
}
template <int NB>
__device__ __forceinline__ typename DVecN<NB>::type dv_load(const double* lrow, int q) {
  return *reinterpret_cast<const typename DVecN<NB>::type*>(lrow + q_off<NB>(q));
}
template <int NB>
__device__ __forceinline__ void dv_store(double* lrow, int q, const typename DVecN<NB>::type& v) {
  *reinterpret_cast<typename DVecN<NB>::type*>(lrow + q_off<NB>(q)) = v;
}

// stored dE/dX (float64 rows holding float32 values) -> accumulator tile
template <int NB>
__device__ __forceinline__ void tile_load_narrow(const double* lrow, Tile<NB>& t) {
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const typename DVecN<NB>::type v = dv_load<NB>(lrow, q);
#pragma unroll
    for (int r = 0; r < NB; ++r) t.b[r][q] = (float)dget<NB>(v, r);
  }
}

constexpr int kPark = 8;   // momentum elements per lane that sit out the GEMM loops in LDS (see pot64_trajectory)
template <int NB>
struct Shared64 {
  Shared<NB> s;
  double red64[4][kP];
  double zn[128 * NB];   // the float64 standard normals of one refreshing column (column_normals)
  f32x4 park[kPark / 2][256];
};

// The wave's working copy of its X and V elements between the passes of a trajectory, in the workgroup's scratch
// (2 x 32 x DIM float64): LANE-LINEAR pieces [wave][array][piece][lane] of 16 bytes (8 at NB = 1), so every access of the
// per-step passes is 64 lanes x 16 B = 1 KB contiguous.  In the particle-major rows themselves a wave instruction
// touches 64 different 128-byte lines (lane = particle, 4 KB apart): the address pipe takes ~64 cycles per instruction
// and, with 128 of them per wave and step, a first form that integrated in the rows spent 40 000 cycles per step there
// (C3 21.3 ms instead of 16.6).  Rows are touched once per trajectory: read by the first pass, written by the last drift
// (X) and the closing kick (V).
// Accesses go through a buffer resource: scalar base + the lane's constant offset + a scalar piece offset.
template <int NB>
struct Work {
  static constexpr int PB = NB == 1 ? 8 : 16;        // piece bytes
  static constexpr int PQ = NB * 8 / PB;             // pieces per register index
  static constexpr unsigned kArea = 16u * PQ * 64u * PB;  // one wave's X (or V) elements: 8192 NB bytes
  __amdgpu_buffer_rsrc_t rs;
  unsigned voff;   // the lane's piece inside the wave's area: w * kArea + lane * PB (vector register)
  static constexpr unsigned xb = 0;   // X areas of the four waves, then (kept for a momentum working copy) the V areas
  static constexpr unsigned vb = 4 * kArea;
};
template <int NB>
__device__ __forceinline__ Work<NB> work_of(double* scratch, unsigned wg, int w, int lane) {
  Work<NB> k;
  char* base = (char*)scratch + (size_t)wg * (8 * Work<NB>::kArea);
  k.rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 8 * Work<NB>::kArea, 0x00020000);
  // (the wave index is threadIdx-derived: a vector value to the compiler.  In the scalar offset of a buffer access it
  // would force a readfirstlane loop around every load and store)
  k.voff = (unsigned)w * Work<NB>::kArea + (unsigned)lane * Work<NB>::PB;
  return k;
}
using uvec4 = __attribute__((ext_vector_type(4))) unsigned;
using uvec2 = __attribute__((ext_vector_type(2))) unsigned;
using f64x2 = __attribute__((ext_vector_type(2))) double;
template <int NB>
__device__ __forceinline__ typename DVecN<NB>::type wk_load(const Work<NB>& k, unsigned area, int q) {
  if constexpr (NB == 1) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(k.rs, k.voff, area + (unsigned)q * 512u, 0));
  } else if constexpr (NB == 2) {
    return __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(k.rs, k.voff, area + (unsigned)q * 1024u, 0));
  } else {
    const f64x2 lo = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(k.rs, k.voff, area + (unsigned)q * 2048u, 0));
    const f64x2 hi = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(k.rs, k.voff, area + (unsigned)q * 2048u + 1024u, 0));
    typename DVecN<4>::type v;
    v[0] = lo[0];
    v[1] = lo[1];
    v[2] = hi[0];
    v[3] = hi[1];
    return v;
  }
}
template <int NB>
__device__ __forceinline__ void wk_store(const Work<NB>& k, unsigned area, int q, const typename DVecN<NB>::type& v) {
  if constexpr (NB == 1) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uvec2, v), k.rs, k.voff, area + (unsigned)q * 512u, 0);
  } else if constexpr (NB == 2) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec4, v), k.rs, k.voff, area + (unsigned)q * 1024u, 0);
  } else {
    f64x2 lo, hi;
    lo[0] = v[0];
    lo[1] = v[1];
    hi[0] = v[2];
    hi[1] = v[3];
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec4, lo), k.rs, k.voff, area + (unsigned)q * 2048u, 0);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uvec4, hi), k.rs, k.voff, area + (unsigned)q * 2048u + 1024u, 0);
  }
}

constexpr int kXWork = 0, kXRows = 1;  // where a drift's X goes: the working copy, the rows xout

// the wave's momentum elements, float64, in registers for the whole trajectory (accumulator layout: 2 x 16 NB registers)
template <int NB>
struct VTile {
  double b[NB][16];
};

// One kick / drift pass over this wave's elements of the tile.  The momentum stays in registers; the position is
// streamed through registers in four chunks of four register indices (double buffered: chunk n + 1 is in flight while
// chunk n is worked on).
//   NKICK = 1: v = [-]V_in + c g                    (the first step of a trajectory; neg: the F of F L F)
//   NKICK = 2: v = (v + c g) + c g                  (closing half kick of a step, opening one of the next)
//   x = X + eps v ;  store x ;  float32(x) -> the X image of GEMM 1
// FIRST: X, V come from the particle rows (the trajectory's first pass), else X from the working copy.  X goes to XDST
// (the last drift's X is the end point: nothing reads it again but the caller).
// Every product is rounded before its sum (the library is built with -ffp-contract=off): NumPy's V += c * g.
template <int NB, int NKICK, bool FIRST, int XDST>
__device__ __forceinline__ void kick_drift_pass(const Work<NB>& wk, const double* xin, const double* vin, double* xout,
                                                const Tile<NB>& g, VTile<NB>& v, double c, double eps, PubWave<NB>* pub0,
                                                int w, int lane, bool neg = false) {
  using DV = typename DVecN<NB>::type;
  DV xa[4], xb[4], va[4], vb[4];   // (va, vb: the first pass only)
  auto load4 = [&](int q4, DV(&x)[4], DV(&vv)[4]) {
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      if constexpr (FIRST) {
        vv[qq] = dv_load<NB>(vin, 4 * q4 + qq);
        x[qq] = dv_load<NB>(xin, 4 * q4 + qq);
      } else {
        x[qq] = wk_load<NB>(wk, wk.xb, 4 * q4 + qq);
      }
    }
  };
  // the momentum first: it needs no memory (the first pass: only its own rows), so it runs while the X loads are in flight
  auto kick4 = [&](int q4, DV(&vin4)[4]) {
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int q = 4 * q4 + qq;
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        const double t = c * (double)g.b[r][q];
        double vv;
        if constexpr (FIRST) {
          vv = dget<NB>(vin4[qq], r);
          vv = neg ? -vv : vv;   // (uniform over the workgroup: an inverse-L item of the jump launch)
        } else {
          vv = v.b[r][q];
        }
        vv = vv + t;
        if constexpr (NKICK == 2) vv = vv + t;
        v.b[r][q] = vv;
      }
    }
  };
  auto drift4 = [&](int q4, DV(&x)[4]) {
    f32x4 px[NB];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int q = 4 * q4 + qq;
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        const double xx = dget<NB>(x[qq], r) + eps * v.b[r][q];
        dset<NB>(x[qq], r, xx);
        px[r][qq] = (float)xx;
      }
      if constexpr (XDST == kXWork) wk_store<NB>(wk, wk.xb, q, x[qq]);
      else if constexpr (XDST == kXRows) dv_store<NB>(xout, q, x[qq]);
    }
    if constexpr (NB == 4) {
      // this lane's slot of the X image, formed from the working copy's lane offset (which every load and store of the
      // pass holds in a register anyway): voff = w * 32 KB + lane * 16 -> w * 16 KB + lane * 16.  As a loop-invariant
      // address it was one more value held -- in scratch -- across the GEMM loops, and its reload in front of the first
      // publish of a pass stood behind an s_waitcnt vmcnt(0) that drained the pass's loads.
      unsigned off = ((wk.voff >> 1) & 0xFFFFC000u) | (wk.voff & 0x3FFu);
      asm volatile("" : "+v"(off));
      char* img = reinterpret_cast<char*>(pub0) + off;
#pragma unroll
      for (int r = 0; r < NB; ++r) *reinterpret_cast<f32x4*>(img + (size_t)(r * 4 + q4) * 1024) = px[r];
    } else {
#pragma unroll
      for (int r = 0; r < NB; ++r) pub0[w].v[r][q4][lane] = px[r];
    }
  };
  // (Measured: all four kick4 first, in the shadow of the first X loads, then the drifts -- more registers live across
  // the pass, 119 -> 283 spilled, C3 in this arithmetic 17.6 -> 18.7 ms.)
  load4(0, xa, va);
  load4(1, xb, vb);
  __builtin_amdgcn_sched_barrier(0);
  kick4(0, va);
  drift4(0, xa);
  __builtin_amdgcn_sched_barrier(0);
  load4(2, xa, va);
  __builtin_amdgcn_sched_barrier(0);
  kick4(1, vb);
  drift4(1, xb);
  __builtin_amdgcn_sched_barrier(0);
  load4(3, xb, vb);
  __builtin_amdgcn_sched_barrier(0);
  kick4(2, va);
  drift4(2, xa);
  __builtin_amdgcn_sched_barrier(0);
  kick4(3, vb);
  drift4(3, xb);
}

// The closing half kick of the trajectory, v += c g, and this lane's part of sum(v^2)
template <int NB>
__device__ __forceinline__ double closing_kick(const Tile<NB>& g, VTile<NB>& v, double c) {
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q)
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      const double vv = v.b[r][q] + c * (double)g.b[r][q];
      v.b[r][q] = vv;
      s = s + vv * vv;
    }
  return s;
}

// sum over the tile's 4 waves x 2 lane halves of a per-lane partial; every lane of column c gets the total.
// One barrier pair.
template <int NB, class SH>
__device__ __forceinline__ double column_sum(SH& sh, int w, int c, int h, double part) {
  const double both = swap32_sum(part);
  if (h == 0) sh.red64[w][c] = both;
  __syncthreads();
  const double tot = sh.red64[0][c] + sh.red64[1][c] + sh.red64[2][c] + sh.red64[3][c];
  __syncthreads();
  return tot;
}

// L >= 1 leapfrog steps (hmc_state.py:86-100) from the rows (xin, vin) and the stored dE/dX in g.  On return the end
// point's position is in the rows xout (XLAST = kXRows; kXNone: not wanted; neg: start from -V, the F of F L F), its momentum in v, g holds its dE/dX (float32), *ex its
// energy (float32, from the last gradient's u), and the return value is its kinetic energy sum(V^2) / 2 (all lanes of
// column c).
template <int NB, int XLAST, class XOut>
__device__ __forceinline__ double pot64_trajectory(const PotModel& mdl, AReg<NB>& ar, Shared64<NB>& sh, int w, int c, int h,
                                                   int lane, const Work<NB>& wk, const double* xin, const double* vin,
                                                   const XOut& xout_of, Tile<NB>& g, VTile<NB>& v, int L, double eps, double chalf,
                                                   float* ex, bool neg) {
  // xout_of(): the rows that take the end point's position, formed only where the last drift needs them (an address
  // held across the GEMM loops and the streamed passes is two registers the kernel does not have)
  PubWave<NB>* pub0 = sh.s.pub[0];   // the X images of the four waves (this wave's: pub0[w])
  if (L == 1) kick_drift_pass<NB, 1, true, XLAST>(wk, xin, vin, xout_of(), g, v, chalf, eps, pub0, w, lane, neg);
  else kick_drift_pass<NB, 1, true, kXWork>(wk, xin, vin, nullptr, g, v, chalf, eps, pub0, w, lane, neg);
  for (int s = 0; s < L; ++s) {
    [[maybe_unused]] const int stamp_slot = s;
    POT_STAMP(0);
    // The momentum (32 NB registers) is not touched by the gradient, whose GEMM loops run at the register budget: the
    // allocator spills a few of its elements across them and reloads them inside the streamed pass, where a scratch
    // reload is a counted load whose wait drains the pass's own loads.  A few elements sit the gradient out in LDS
    // instead (lane-linear, conflict-free; an LDS wait drains nothing).
    if constexpr (NB == 4) {
#pragma unroll
      for (int k = 0; k < kPark / 2; ++k) {
        f64x2 two;
        two[0] = v.b[NB - 1][15 - 2 * k];
        two[1] = v.b[NB - 1][14 - 2 * k];
        sh.park[k][threadIdx.x] = __builtin_bit_cast(f32x4, two);
      }
    }
    pot_gradient_published<NB>(mdl, ar, sh.s, w, c, h, lane, g, s == L - 1, ex, s);
    if constexpr (NB == 4) {
#pragma unroll
      for (int k = 0; k < kPark / 2; ++k) {
        const f64x2 two = __builtin_bit_cast(f64x2, sh.park[k][threadIdx.x]);
        v.b[NB - 1][15 - 2 * k] = two[0];
        v.b[NB - 1][14 - 2 * k] = two[1];
      }
      // (whatever the allocator still keeps of the momentum in scratch across the GEMM loops comes back HERE, in front of
      // the pass's loads, not between them)
#pragma unroll
      for (int r = 0; r < NB; ++r)
#pragma unroll
        for (int q = 0; q < 16; ++q) use_here(v.b[r][q]);
    }
    POT_STAMP(6);
    if (s < L - 2) kick_drift_pass<NB, 2, false, kXWork>(wk, nullptr, nullptr, nullptr, g, v, chalf, eps, pub0, w, lane);
    else if (s == L - 2) kick_drift_pass<NB, 2, false, XLAST>(wk, nullptr, nullptr, xout_of(), g, v, chalf, eps, pub0, w, lane);
    POT_STAMP(7);
  }
  const double part = closing_kick<NB>(g, v, chalf);
  return column_sum<NB>(sh, w, c, h, part) / 2.0;
}

// ---------------------------------------------------------------------------------------------------
// The inverse-L proposal: which particles integrate it (not the F-movers), its tiles as items of the jump launch, the
// pending particles and the fix kernel -- dense_pot.hip has the story; this is the float64-state twin.
// ---------------------------------------------------------------------------------------------------
__global__ void pot64_cold_list_kernel(const double* __restrict__ Hflf_in, const double* __restrict__ Hspec_in, int64_t N,
                                       int* __restrict__ list, int* __restrict__ count, const Control* ctl) {
  if (ctl->failed) return;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const double hc = p < N ? Hflf_in[p] : 0.0, hs = p < N ? Hspec_in[p] : 0.0;
  append_cold(list, count, (p < N) && !(hc == hc) && !(hs == hs), p);
}

// what the kernel that only DECIDES and finishes the moves needs of Shared64 (pot64_decide_kernel)
template <int NB>
struct Finish64Shared {
  struct {
    int move[kP];
  } s;
  double red64[4][kP];
  double zn[128 * NB];
};

// The successor's rows once the moves of a tile's columns stand in sh.s.move.  FIX = false (jump kernel): the end point's
// position is in the output rows, its momentum in v, its dE/dX in g.  FIX = true (pot64_decide_kernel): columns that keep the
// end point are finished already (v, g unused); only the others are touched.  roff: this lane's elements of its column's row.
template <int NB, bool REPLAY, int MODE, bool FIX, class SH>
__device__ __forceinline__ void pot64_finish(const Pot64JumpArgs& a, SH& sh, int64_t p, bool alive, size_t roff, int w, int c,
                                             int h, const VTile<NB>& v, const Tile<NB>& g) {
  using DV = typename DVecN<NB>::type;
  const double* xin = a.X_in + roff;
  const double* vin = a.V_in + roff;
  const double* gin = a.G_in + roff;
  double* xo = a.X_out + roff;
  double* vo = a.V_out + roff;
  double* go = a.G_out + roff;
  const int mv = sh.s.move[c];
  const int k = mv & 3;
  bool take, flip, refresh;  // keep the end point of L; negate the successor's momentum; redraw it (HMCState.R)
  if constexpr (MODE == kModeControl) {
    const bool accept = k & 1, fl = k & 2;
    take = accept;
    flip = accept != fl;      // accepted: L F, then possibly F again; rejected: possibly F (markov_jump_hmc.py:116-141)
    refresh = (mv & 4) != 0;  // batch-wide (:138-141)
  } else {
    take = k == 0;
    flip = (MODE == kModeCT && k == 0) || k == 1;  // CT's FL move ends with a flip (:258,278); F flips
    refresh = k == 2;
  }
  // the successor's rows.  A kept end point: position already in the output rows, momentum from the registers
  // (negated where the move ends with a flip), dE/dX from the accumulator; everything else: the pre-move position and
  // dE/dX, and in the second loop its momentum, flipped / refreshed
  const bool tile_refreshes = __ballot(refresh) != 0ull;
  if (!FIX || !take) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      DV gg;
      if (take) {
        if constexpr (!FIX) {
          DV vv;
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            dset<NB>(vv, r, flip ? -v.b[r][q] : v.b[r][q]);
            dset<NB>(gg, r, (double)g.b[r][q]);
          }
          dv_store<NB>(vo, q, vv);
        }
      } else {
        gg = dv_load<NB>(gin, q);
        dv_store<NB>(xo, q, dv_load<NB>(xin, q));
      }
      dv_store<NB>(go, q, gg);
    }
  }
  double s2 = 0.0;
  if (!take || (REPLAY && refresh)) {
    const double* vsrc = take ? (const double*)vo : vin;   // (a kept end point's momentum is flipped already)
    const bool fl2 = flip && !take;
    const double* nrow = REPLAY ? lane_row<NB>(a.noise, alive ? p : 0, w, h) : nullptr;
#pragma unroll 1
    for (int q = 0; q < 16; ++q) {
      DV vv = dv_load<NB>(vsrc, q);
      if (fl2) vv = -vv;
      if constexpr (REPLAY) {
        if (refresh) {  // HMCState.R (hmc_state.py:121-129) with the recorded normals
          const DV z = dv_load<NB>(nrow, q);
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            const double t = dget<NB>(vv, r) * a.r_keep + dget<NB>(z, r) * a.r_mix;
            dset<NB>(vv, r, t);
            s2 = s2 + t * t;
          }
        }
      }
      dv_store<NB>(vo, q, vv);
    }
  }
  if constexpr (!REPLAY) {
    // HMCState.R, column by column (the set is the same in every wave: it comes from sh.move), the whole workgroup
    // drawing the column's normals (column_normals); the column's momentum is in the output rows by now
    unsigned cols = (unsigned)(__ballot(refresh) & 0xFFFFFFFFull);
    while (cols) {
      const int c0 = __ffs((int)cols) - 1;
      cols &= cols - 1;
      const int64_t p0 = __shfl((long long)p, c0);
      column_normals<NB, double>(a.key, (uint32_t)(a.first_pid + (p0 < a.N ? p0 : 0)), a.D, sh.zn);
      __syncthreads();
      if (c == c0) {
        const double* zrow = sh.zn + 32 * NB * w + 4 * NB * h;
#pragma unroll 1
        for (int q = 0; q < 16; ++q) {
          DV vv = dv_load<NB>(vo, q);
          const DV z = *reinterpret_cast<const DV*>(zrow + q_off<NB>(q));
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            const double t = dget<NB>(vv, r) * a.r_keep + dget<NB>(z, r) * a.r_mix;
            dset<NB>(vv, r, t);
            s2 = s2 + t * t;
          }
          dv_store<NB>(vo, q, vv);
        }
      }
      __syncthreads();
    }
  }
  if (tile_refreshes) {  // all waves take part in the reduction; only refreshed columns use the result
    const double evr = column_sum<NB>(sh, w, c, h, s2) / 2.0;
    if (refresh && w == 0 && h == 0) a.EV_out[p] = evr;
  }
}

// rates / acceptance, waiting times, first minimum of ONE particle by one lane, with the device functions of the
// elementwise kernels in their one-lane-per-particle forms (as hk_decide of the multi-pass path)
template <bool REPLAY, int MODE>
__device__ __forceinline__ int pot64_decide(const Pot64JumpArgs& a, double H0, double HL, double Hflf, int64_t pp, uint32_t pid,
                                            double& best, bool& bad, bool& gate) {
  JumpArgs<double> ja;
  ja.p_r = a.p_r;
  ja.p_flip = a.p_flip;
  ja.rexp = a.rexp;
  ja.runif = a.runif;
  ja.N = a.N;
  LaneMap m;
  m.j = 0;
  m.G = 1;
  m.D = 1;
  m.CH = 1;
  m.lane0 = 0;
  m.wpp = 0;
  int k = 0;
  if constexpr (MODE == kModeMJHMC) {
    decide<double, REPLAY>(ja, a.key, m, H0, HL, Hflf, pp, pid, k, best, bad);
  } else if constexpr (MODE == kModeCT) {
    decide_ct<double, REPLAY>(ja, a.key, m, H0, HL, pp, pid, k, best, bad);
  } else {
    double uacc, uflip, ugate;
    if constexpr (REPLAY) {
      uacc = a.runif[pp];
      uflip = a.runif[a.N + pp];
      ugate = a.runif[2 * a.N];
    } else {
      const u32x4 q = philox4x32_10(pid, a.key.tick_lo, a.key.tick_hi, kSlotExpR, a.key.k0, a.key.k1);
      const u32x4 f = philox4x32_10(pid, a.key.tick_lo, a.key.tick_hi, kSlotFlip, a.key.k0, a.key.k1);
      const u32x4 gq = philox4x32_10(0xFFFFFFFFu, a.key.tick_lo, a.key.tick_hi, kSlotFlip, a.key.k0, a.key.k1);
      uacc = u53(q.w2, q.w3);
      uflip = u53(f.w0, f.w1);
      ugate = u53(gq.w2, gq.w3);
    }
    const double dH = H0 - HL;
    const bool accept = !(dH < 0.0) || (uacc < exp(dH));
    const bool flip = uflip < a.p_flip;
    gate = ugate < a.p_r;
    k = (accept ? 1 : 0) | (flip ? 2 : 0);
  }
  return k;
}

// ---------------------------------------------------------------------------------------------------
// the jump kernel: one sampling_iteration attempt for a tile of 32 particles (MODE as in pot_jump_kernel)
// ---------------------------------------------------------------------------------------------------
template <int NB, bool REPLAY, int MODE>
__global__ __launch_bounds__(256, 1) void pot64_jump_kernel(const Pot64JumpArgs a, const PotModel mdl) {
  __shared__ Shared64<NB> sh;
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
  // tallies (meaning per mode: fill_iter_stats in api.hip) and the failure flag live in LDS, not in registers that would be
  // live across every GEMM loop and streamed pass of the kernel: [0..3] counts, [4] some particle met a non-finite rate
  __shared__ unsigned tally[5];
  if (threadIdx.x < 5) tally[threadIdx.x] = 0;
  AReg<NB> ar;
  areg_load<NB>(mdl, w, c, h, ar);
  stage_bias<NB>(mdl, sh.s);
  const Work<NB> wk = work_of<NB>(a.scratch, blockIdx.x, w, lane);
  for (int64_t item = blockIdx.x; item < nft + a.ntiles; item += gridDim.x) {
    const bool inverse = item < nft;   // (uniform over the workgroup)
    auto column_of = [&](int64_t it) -> int64_t {
      if (it < nft) {
        const int64_t slot = it * kP + c;
        return a.cold_list[slot < ncold ? slot : ncold - 1];  // pad the last tile with a repeat
      }
      return (it - nft) * kP + c;
    };
    // (everything that is not needed during the trajectory -- the scalars of the decision, the output rows' addresses --
    // is fetched / formed after it: the kernel sits at its 512-register budget, and what is live across the GEMM loops
    // and the streamed passes decides whether those spill; tools/check_isa.sh gates both.  That includes the column's own
    // index: it is looked up again behind the trajectory, through an item number the compiler cannot see through)
    Tile<NB> g;
    VTile<NB> v;
    float exl = 0.f;
    double EVL;
    {
      const int64_t p = column_of(item);
      const size_t roff = (size_t)p * (128 * NB) + 32 * NB * w + 4 * NB * h;   // this lane's elements inside a [*][DIM] matrix
      // the end point's position: into the output rows -- an inverse-L item's is wanted by nobody: into the second half of
      // the workgroup's working rows (32 rows, the momentum's would-be working copy: unused)
      auto xend = [&]() -> double* {
        int64_t it = item;
        asm volatile("" : "+s"(it));
        const size_t lane_part = 32 * NB * w + 4 * NB * h;
        return it < nft ? a.scratch + (size_t)blockIdx.x * (Work<NB>::kArea) + Work<NB>::kArea / 2 + (size_t)c * (128 * NB) + lane_part
                        : a.X_out + (size_t)column_of(it) * (128 * NB) + lane_part;
      };
      tile_load_narrow<NB>(a.G_in + roff, g);
      EVL = pot64_trajectory<NB, kXRows>(mdl, ar, sh, w, c, h, lane, wk, a.X_in + roff, a.V_in + roff, xend, g, v, a.L, a.eps,
                                         a.chalf, &exl, inverse);
    }
    int64_t item_again = item;
    asm volatile("" : "+s"(item_again));
    const int64_t p = column_of(item_again);
    const bool alive = p < a.N;
    const size_t roff = (size_t)p * (128 * NB) + 32 * NB * w + 4 * NB * h;
    const double EXL = (double)exl;
    const double HL = EXL + EVL;
    if (inverse) {
      if (w == 0 && h == 0) a.Hwork[p] = HL;
      __syncthreads();
      continue;
    }

    if constexpr (MODE == kModeMJHMC) {
      // the jump process itself -- rates, clocks, first minimum, the successor of a move that is not L -- belongs to
      // pot64_decide_kernel, which runs when this launch's inverse-L items are done too: here the end point of L is
      // written as if taken (position: by the last drift), with its energies
      if (w == 0 && h == 0) {
        a.EX_out[p] = EXL;
        a.EV_out[p] = EVL;
      }
      using DV = typename DVecN<NB>::type;
      double* vo = a.V_out + roff;
      double* go = a.G_out + roff;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        DV vv, gg;
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          dset<NB>(vv, r, v.b[r][q]);
          dset<NB>(gg, r, (double)g.b[r][q]);
        }
        dv_store<NB>(vo, q, vv);
        dv_store<NB>(go, q, gg);
      }
      __syncthreads();
      continue;
    }

    // the discrete-time and continuous-time control samplers decide here: lanes 0..31 of wave 0, one particle each
    if (w == 0 && h == 0) {
      const int64_t pp = alive ? p : 0;
      const uint32_t pid = (uint32_t)(a.first_pid + pp);
      const double EX0 = a.EX_in[p], EV0 = a.EV_in[p];
      const double H0 = EX0 + EV0;
      double best = 0.0;
      bool bad = false, gate = false;
      const int k = pot64_decide<REPLAY, MODE>(a, H0, HL, 0.0, pp, pid, best, bad, gate);
      if (bad && alive) tally[4] = 1;
      a.dwell[p] = best;
      a.dwell_ring[p] = best;
      a.trans[p] = (uint8_t)k;
      sh.s.move[c] = k | (gate ? 4 : 0);
      {
        // one LDS atomic per tally and tile (the 32 deciding lanes of wave 0)
        unsigned long long b0, b1, b2, b3 = 0ull;
        if constexpr (MODE == kModeControl) {  // l_count, f_count, R applied, fl_count (markov_jump_hmc.py:143-148)
          b0 = __ballot(alive && k == 3);
          b1 = __ballot(alive && k == 2);
          b2 = __ballot(alive && gate);
          b3 = __ballot(alive && k == 1);
        } else {
          b0 = __ballot(alive && k == 0);
          b1 = __ballot(alive && k == 1);
          b2 = __ballot(alive && k == 2);
        }
        if (c == 0) {
          if (b0) atomicAdd(&tally[0], (unsigned)__popcll(b0));
          if (b1) atomicAdd(&tally[1], (unsigned)__popcll(b1));
          if (b2) atomicAdd(&tally[2], (unsigned)__popcll(b2));
          if (b3) atomicAdd(&tally[3], (unsigned)__popcll(b3));
        }
      }
      // scalars of the successors that keep or take whole states; a refreshed kinetic energy is filled in below
      const bool took_L = MODE == kModeControl ? (k & 1) : (k == 0);
      a.EX_out[p] = took_L ? EXL : EX0;
      a.EV_out[p] = took_L ? EVL : EV0;
      a.Hflf_out[p] = __builtin_nan("");
    }
    __syncthreads();
    pot64_finish<NB, REPLAY, MODE, false>(a, sh, p, alive, roff, w, c, h, v, g);
    __syncthreads();
  }
  __syncthreads();
  if (threadIdx.x == 0 && tally[4]) {
    a.ctl->failed = 1;
    a.ctl->failed_iter = a.iter;
  }
  if (threadIdx.x < 4 && tally[threadIdx.x]) atomicAdd(&a.stats[threadIdx.x], (unsigned long long)tally[threadIdx.x]);
}

// MarkovJumpHMC's jump process for every particle (markov_jump_hmc.py:366-415), once both trajectories of the iteration
// are done -- the L proposal in the output rows with its energies in EX_out / EV_out, H of the inverse-L proposal cached
// (H_flf), handed on by the F move of the iteration before (Hspec) or integrated by this iteration's launch (Hwork):
// rates, clocks, first minimum, counters, the next iteration's list, and where the move is not L the pre-move position
// and dE/dX put back and the momentum flipped / redrawn (pot_decide_kernel's twin).
template <int NB, bool REPLAY>
__global__ __launch_bounds__(256) void pot64_decide_kernel(const Pot64JumpArgs a) {
  __shared__ Finish64Shared<NB> sh;
  if (a.ctl->failed) return;
  const int w = threadIdx.x >> 6, c = threadIdx.x & 31, h = (threadIdx.x & 63) >> 5;
  unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0;
  bool any_bad = false;
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int64_t p = tile * kP + c;
    const bool alive = p < a.N;
    if (w == 0 && h == 0) {
      const int64_t pp = alive ? p : 0;
      const uint32_t pid = (uint32_t)(a.first_pid + pp);
      const double EX0 = a.EX_in[p], EV0 = a.EV_in[p];
      const double H0 = EX0 + EV0;
      const double HL = a.EX_out[p] + a.EV_out[p];
      double Hflf = a.Hflf_in[p];
      const bool cold = !(Hflf == Hflf);
      if (cold) {
        Hflf = a.Hspec_in[p];                      // the L proposal of the iteration in which the particle flipped ...
        if (!(Hflf == Hflf)) Hflf = a.Hwork[p];    // ... or integrated by an inverse-L item of this iteration's launch
      }
      double best = 0.0;
      bool bad = false, gate = false;
      const int k = pot64_decide<REPLAY, kModeMJHMC>(a, H0, HL, Hflf, pp, pid, best, bad, gate);
      sh.s.move[c] = alive ? k : 0;
      any_bad |= bad && alive;
      // every move but L clears the cache; of those only the R-movers need their inverse-L proposal integrated
      append_cold(a.next_list, a.next_count, alive && k == 2, p);
      a.dwell[p] = best;
      a.dwell_ring[p] = best;
      a.trans[p] = (uint8_t)k;
      if (alive) {
        n0 += (k == 0);
        n1 += (k == 1);
        n2 += (k == 2);
        n3 += cold;   // the reference integrates F L F for every one of these
      }
      if (k != 0) {
        a.EX_out[p] = EX0;
        a.EV_out[p] = EV0;   // (an R-mover's: filled in by pot64_finish)
      }
      a.Hflf_out[p] = k == 0 ? H0 : __builtin_nan("");
      a.Hspec_out[p] = k == 1 ? HL : __builtin_nan("");
    }
    __syncthreads();
    if (__syncthreads_or(sh.s.move[c] != 0)) {   // (a tile of L-movers is finished already)
      const size_t roff = (size_t)p * (128 * NB) + 32 * NB * w + 4 * NB * h;
      VTile<NB> v;   // (unused: FIX)
      Tile<NB> g;
      pot64_finish<NB, REPLAY, kModeMJHMC, true>(a, sh, p, alive, roff, w, c, h, v, g);
      __syncthreads();
    }
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

static int resident_cus64() {
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return std::max(1, cus);
}

int pot64_scratch_workgroups() { return resident_cus64(); }

template <int NB, int MODE>
static void launch64_mode(const Pot64JumpArgs& a, const PotModel& mdl, unsigned grid, hipStream_t st) {
  const bool replay = MODE == kModeControl ? (a.runif && a.noise) : (a.rexp && a.noise);
  if (replay) hipLaunchKernelGGL((pot64_jump_kernel<NB, true, MODE>), dim3(grid), dim3(256), 0, st, a, mdl);
  else hipLaunchKernelGGL((pot64_jump_kernel<NB, false, MODE>), dim3(grid), dim3(256), 0, st, a, mdl);
}

template <int NB>
static void launch64_nb(const Pot64JumpArgs& a, const PotModel& mdl, hipStream_t st) {
  const int cus = resident_cus64();
  if (a.mode == kModeMJHMC) {  // only MJHMC has the inverse-L proposal and its cache
    if (a.iter == 0 || a.rescan) {  // first iteration of a call: the three counters cleared (they are adjacent), the list from a scan
      (void)hipMemsetAsync(std::min(a.cold_count, std::min(a.next_count, a.zero_count)), 0, 3 * sizeof(int), st);
      hipLaunchKernelGGL(pot64_cold_list_kernel, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, st, a.Hflf_in, a.Hspec_in,
                         a.N, a.cold_list, a.cold_count, (const Control*)a.ctl);
    }
    // forward tiles + at most as many inverse-L tiles (workgroups without an item leave at once)
    const unsigned grid = (unsigned)std::min<int64_t>(2 * a.ntiles, cus);
    launch64_mode<NB, kModeMJHMC>(a, mdl, grid, st);
    const unsigned fgrid = (unsigned)std::min<int64_t>(a.ntiles, 4 * cus);
    if (a.rexp && a.noise) hipLaunchKernelGGL((pot64_decide_kernel<NB, true>), dim3(fgrid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((pot64_decide_kernel<NB, false>), dim3(fgrid), dim3(256), 0, st, a);
  } else {
    const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, cus);
    if (a.mode == kModeCT) launch64_mode<NB, kModeCT>(a, mdl, grid, st);
    else launch64_mode<NB, kModeControl>(a, mdl, grid, st);
  }
}

void pot64_launch_jump(const Pot64JumpArgs& a, const PotModel& mdl, hipStream_t st) {
  if (mdl.dim == 128) launch64_nb<1>(a, mdl, st);
  else if (mdl.dim == 256) launch64_nb<2>(a, mdl, st);
  else launch64_nb<4>(a, mdl, st);
}

#ifdef POT_STAMPS
}  // namespace mjhmc
extern "C" int mjhmc_pot64_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mjhmc::g_pot_stamp), sizeof(mjhmc::g_pot_stamp));
}
namespace mjhmc {
#endif

}  // namespace mjhmc
