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
// ROUND 3: ONE pass over the dictionary per leapfrog step.  Both products of a step read the same matrix -- the kick
// needs B^T (rows = coefficients), the next residual needs B (rows = pixels) -- and the dictionary (512 KB in bf16)
// cannot stay in a CU, so a 32-column tile is bound by the rate at which its CU can pull dictionary bytes out of L2
// (measured 36-41 B/clk/CU, 0.6 of the L2 peak, whatever the prefetch depth).  Rounds 1-2 streamed it twice per step.
// Now a block of 32 coefficient rows (all 256 pixels: 16 KB) is loaded ONCE and used for both:
//
//   one workgroup = 8 waves (two per SIMD, 256 registers each) owns a tile of 32 columns; wave w holds coefficient rows
//   [128 w, 128 w + 128) of X and V as fp32 MFMA accumulator tiles (4 blocks of 32 rows) and pixel rows [32 w, 32 w + 32)
//   of the residual (1 block).  The waves form two groups, {0..3} and {4..7} (a SIMD holds one wave of each).  A leapfrog
//   step is 8 ROUNDS; round r belongs to group r & 1, whose four waves each OWN one block of it (block r >> 1 of theirs):
//     owners   G2   V_block += B^T[block] . (s * resid)       16 v_mfma_f32_32x32x16_bf16, A = the block's image read row-wise
//              drift X_block += eps * V_block; publish bf16(X_block) as two B fragments
//     (the other group meanwhile adds the prior's force of ITS next block: vector work beside the owners' matrix work)
//                                                                                                        (barrier A)
//     all 8    G1   resid'[32 w ..] += B[32 w .., the round's four blocks] . X(those blocks):  8 MFMAs per wave,
//              A = the SAME LDS images read TRANSPOSED (ds_read_b64_tr_b16)                            (barrier B)
//     owners   start the LDS-DMA of their next block into the image they have just finished with (a step's pass: the
//              first 12 of its 16 requests -- the SIMD partner makes the other 4 after ITS publish in the next round: the
//              requesting waves stand at the vector-memory port, 64 B/clk per CU, and their path to barrier A was the
//              longer one; DESIGN.md section 3.5 has the cycle accounting)
//   so the kick of step n (residual of step n) and the residual of step n + 1 come out of one stream of the dictionary:
//   L + 2 passes per trajectory instead of 2 L + 2.  Image buffers: 2 groups x 4 owners x 16 KB of LDS, filled by
//   `global_load_lds_dwordx4` (no VGPR destination) while the OTHER group's round runs -- across rounds, passes and tiles.
//   (A first form of this with 4 waves of 512 registers, every wave owning a block in every round, was correct and 1.6x
//   SLOWER than the two-pass kernel: hipcc splits 512 registers into 256 vector + 256 accumulation registers, vector
//   instructions cannot touch the latter, and the kernel spent its time on v_accvgpr moves, spills and exposed LDS
//   latency with nothing else on the SIMD to cover it.)
//
// Accumulator registers 8s..8s+7 converted to bf16 are directly the B fragment of k-step s (rows 16s + 8(j>>2) + 4h +
// (j&3)); the dictionary is pre-permuted on the host into that k-order ("A2": [k-step of 16 pixels][half][coefficient]
// [8 pixels]).  Inside an LDS image the 64 16-byte chunks of a k-step are PERMUTED (the DMA's per-lane source offset
// does it for free) so that the row read of G2 (ds_read_b128) and the transposed read of G1 are both bank-conflict free.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "dense_sic.hpp"
#include "timing_variants.hpp"   // SIC_STAMP: cycle stamps of the timing build (tools/sic_variants.sh); nothing otherwise

namespace mjhmc {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using i16x4 = __attribute__((ext_vector_type(4))) short;

constexpr int kP = 32;
constexpr int kI = kSicImg;     // 256
constexpr int kW = 8;           // waves per workgroup, two per SIMD
constexpr int kG = 4;           // waves per group = blocks per round
// NB = 32-row blocks of X and V per wave = n_coeffs / 256 (4: the 1024-atom dictionary, 2: the 512-atom one)
#define kC (256 * NB)

__device__ __forceinline__ int acc_row(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

template <int NB>
struct CTile {
  f32x16 b[NB];  // this wave's 32 NB coefficient rows x 32 columns
};
struct RTile {
  f32x16 b[1];    // this wave's 32 pixel rows of the residual
};

// bf16 state rows [*][n_coeffs]: lane (c, h) reads its 16 groups of 4 consecutive coefficients (8 bytes each)
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

// float32 state rows (the reference's own state type): the same elements, 16 bytes per group of 4
template <int NB>
__device__ __forceinline__ void ctile_load(const float* base, int64_t p, int w, int h, CTile<NB>& t) {
  const float* row = base + (size_t)p * kC + 32 * NB * w + 4 * h;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(row + 32 * b + 8 * g);
#pragma unroll
      for (int k = 0; k < 4; ++k) t.b[b][4 * g + k] = v[k];
    }
}
template <int NB>
__device__ __forceinline__ void ctile_store_f32(float* base, int64_t p, int w, int h, const CTile<NB>& t);
template <int NB>
__device__ __forceinline__ void ctile_store(float* base, int64_t p, int w, int h, const CTile<NB>& t) {
  ctile_store_f32(base, p, w, h, t);
}

// dE/dX is handed out in float32
template <int NB>
__device__ __forceinline__ void ctile_store_f32(float* base, int64_t p, int w, int h, const CTile<NB>& t) {
  float* row = base + (size_t)p * kC + 32 * NB * w + 4 * h;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 o;
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = t.b[b][4 * g + k];
      *reinterpret_cast<f32x4*>(row + 32 * b + 8 * g) = o;
    }
}

constexpr int kYLds = 4;    // patches whose pixels fit the LDS copy (more: read from global memory at the head of a pass)

// ALL of the workgroup's LDS is this one object (a second __shared__ object beside an LDS-DMA target makes hipcc
// drain the DMA queue before LDS reads, cdna_hip_programming.md section 5)
struct SicShared {
  f32x4 img[2][kG][16][64];   // 128 KB: dictionary block images, [owner group = round parity][owner][k-step][chunk position]
  f32x4 pubR[16][64];         // 16 KB: the scaled residual as B fragments [k-step][lane], written at the head of a pass
  f32x4 pubX[kG][2][64];      //  8 KB: a round's drifted coefficient blocks as B fragments [wave][k-step][lane]
  float ys[kYLds * kI];       //  4 KB: the patches
  int ypatch[kP];             // byte offset of column c's patch inside ys (the column -> patch map does not depend on the tile)
  float red[2][kW][kP];
  float colsum[kP];           // per-column energies, summed over a particle's columns when n_patches > 1
  int move[kP];
  unsigned tally[4];
#ifdef SIC_STAMPS
  unsigned stamp[4][8][8];    // timing build only
#endif
};
static_assert(sizeof(SicShared) <= 160 * 1024, "LDS budget of a CU");

// ---- the dictionary stream -----------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}
// lane l's 16 bytes at (sbase + voff) land at lds_base + LDS_OFF + 16 l.  sbase and lds_base are wave-uniform (scalar
// registers), voff the lane's byte offset; the DMA is inline asm, so hipcc neither counts it nor drains it at a barrier.
// Round 6: THREE instructions per request instead of nine.  The request used to be `save m0; m0 = dst; nop; load; restore m0`
// behind TWO v_readlane: every (block, k-step) source address was its own loop-invariant scalar pair -- 128 of them per
// leapfrog step, 329 scalar registers spilled into vector lanes and read back one request at a time.  Now the k-step's and
// the block's offset ride in the lane's VECTOR offset (one v_add with a literal), the source base is ONE scalar pair per
// wave, the destination is `s_add_u32 m0, base, literal`, and m0 is declared clobbered instead of saved and restored
// (nothing else in these kernels reads it).
template <unsigned LDS_OFF, unsigned SRC_OFF>
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_base) {
  unsigned at;   // (the add stands INSIDE the statement, between the write of m0 and its use -- the wait state the pair needs --:
                 // left to the compiler the 112 sums of a step were formed ahead and kept in registers the kernel does not have)
  asm volatile("s_add_u32 m0, %3, %4\n\tv_add_u32 %0, %5, %1\n\tglobal_load_lds_dwordx4 %0, %2"
               : "=&v"(at)
               : "v"(voff), "s"(sbase), "s"(lds_base), "i"(LDS_OFF), "i"(SRC_OFF)
               : "memory", "m0", "scc");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

#define kFrag2 (2u * kC * 16u)   // bytes between consecutive k-steps of A2; the blocks of a k-step are 512 B apart

// Chunk position inside a k-step image.  A k-step image holds, for 16 pixels, 64 chunks of 16 bytes: chunk (hs, c) =
// the 8 pixels {8(j>>2) + 4 hs + (j&3)} of coefficient c (0..31 inside the block).  It sits at position
//     pos = 32 hs + 16 (c >> 4) + 4 (((c >> 2) & 3) ^ (hs + 2 (ks & 1))) + (c & 3)
// (16-byte units).  Row read of G2, lane (c, h) takes chunk (h, c): inside each of ds_read_b128's 16-lane groups the four
// values of (c >> 2) & 3 are distinct, XOR-ing a constant keeps them so.  Transposed read of G1: a 32-lane half takes, for
// 4 consecutive coefficients, both pixel halves of two consecutive k-steps -- (c >> 2) & 3 is fixed there and the XOR
// term spreads (hs, ks & 1) over the four 64-byte bank groups.  Both reads are conflict free.
__device__ __forceinline__ int chunk_pos(int hs, int c, int ks_odd) {
  return 32 * hs + 16 * (c >> 4) + 4 * (((c >> 2) & 3) ^ (hs + 2 * ks_odd)) + (c & 3);
}

struct AStream {
  const char* a2w;     // A2 (+ copy) at this wave's first block: wave-uniform
  const char* a2p;     // ... and at its SIMD partner's (wave w ^ 4)
  unsigned img_partner;   // LDS byte address of the partner's image buffer
  unsigned voff[2];    // byte offset, inside a k-step of A2, of the chunk that lands at THIS lane's position (even / odd k-step)
  unsigned img_own;    // LDS byte address of this wave's image buffer, img[w >> 2][w & 3]
  unsigned row[2];     // byte offset of this lane's G2 row-read chunk inside a k-step image (even / odd k-step)
  unsigned tr[2];      // byte offset of this lane's G1 transposed reads inside an owner's block image, first / second half of
                       // the fragment, at k-step 2 w + (lane group & 1) and k-step-of-coefficients 0
};

// k-steps [K0, K1) of block T (compile time) of this wave into its image buffer
template <int NB, int T, int KS, int K1>
__device__ __forceinline__ void issue_ks(const char* base, const unsigned (&voff)[2], unsigned img) {
  if constexpr (KS < K1) {
    glds16<(unsigned)KS * 1024u, (unsigned)((T % NB) * 512 + KS * (int)kFrag2)>(base, voff[KS & 1], img);
    issue_ks<NB, T, KS + 1, K1>(base, voff, img);
  }
}
template <int NB, int T, int K0 = 0, int K1 = 16>
__device__ __forceinline__ void issue_block(const AStream& s) {
  issue_ks<NB, T, K0, K1>(s.a2w, s.voff, s.img_own);
}
// the same for the wave's SIMD partner (wave w ^ 4, the other group): DIR = +1 from group 0, -1 from group 1
template <int NB, int T, int K0, int K1, int DIR>
__device__ __forceinline__ void issue_partner(const AStream& s) {
  issue_ks<NB, T, K0, K1>(s.a2p, s.voff, s.img_partner);
}
// A step's pass: of the sixteen requests of an owner's next block the last 16 - kOwnPieces are made by its SIMD partner.
// The waves that request stand at the vector-memory port (64 B/clk per CU: ~58 cycles per request with four waves
// asking) and then run the prior's force: ~1300 cycles to barrier A where this round's owners need ~1000 (G2, drift,
// publish) and wait.  The owners take over the tail of the burst after their publish.
constexpr int kOwnPieces = 12;   // (measured 0..16: DESIGN.md section 3.5)

template <int NB>
__device__ __forceinline__ AStream astream_open(const SicModel& mdl, SicShared& sh, int w, int lane) {
  AStream s;
  const size_t copy = (size_t)((blockIdx.x >> 3) % (unsigned)mdl.copies) * (size_t)(kI * kC * 2);  // blocks b, b + 8 share an XCD
  s.a2w = reinterpret_cast<const char*>(mdl.A2) + copy + (size_t)(32 * NB * w) * 16;
  // (the pointer is wave-uniform; read through the first lane so that it lives in scalar registers)
  s.a2w = (const char*)(((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long)s.a2w >> 32)) << 32) |
                        (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long)s.a2w));
  s.img_own = __builtin_amdgcn_readfirstlane(lds_addr(&sh.img[w >> 2][w & 3][0][0]));
  {
    const long dir = (w < 4) ? 1 : -1;   // the partner of a wave of group 0 is four waves up, of group 1 four down
    s.a2p = s.a2w + dir * (long)(4 * 32 * NB * 16);
    s.a2p = (const char*)(((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((unsigned long)s.a2p >> 32)) << 32) |
                          (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long)s.a2p));
    s.img_partner = __builtin_amdgcn_readfirstlane(lds_addr(&sh.img[(w >> 2) ^ 1][w & 3][0][0]));
  }
  const int c = lane & 31, h = lane >> 5;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    // which chunk lands at position `lane` of an image of k-step parity p (the inverse of chunk_pos)
    const int hs = lane >> 5, x = (lane >> 2) & 3;
    const int cc = 16 * ((lane >> 4) & 1) + 4 * (x ^ (hs + 2 * p)) + (lane & 3);
    s.voff[p] = (unsigned)(hs * kC + cc) * 16u;
    s.row[p] = (unsigned)chunk_pos(h, c, p) * 16u;
  }
  {  // transposed reads: 16-lane group g = lane >> 4 serves MFMA rows 16 (g & 1) .. + 15 of lane half g >> 1
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
    const int hh = g >> 1, ko = g & 1;  // the fragment's lane half; parity of the k-step these 16 pixel rows live in
#pragma unroll
    for (int u = 0; u < 2; ++u)        // coefficients 8 u + 4 hh + q of a 16-coefficient k-step; pixels 4 pp .. 4 pp + 3
      s.tr[u] = (unsigned)chunk_pos(pp & 1, 8 * u + 4 * hh + q, ko) * 16u + (unsigned)(pp >> 1) * 8u + (unsigned)ko * 1024u +
                (unsigned)(2 * w) * 1024u;
  }
  issue_block<NB, 0>(s);   // the head of the stream
  return s;
}

// before the wave exits: no DMA may still be writing LDS that the next workgroup will own
__device__ __forceinline__ void astream_close() { wait_vm<0>(); }

// the patches into LDS (once per kernel)
__device__ __forceinline__ void stage_patches(const SicModel& mdl, SicShared& sh) {
  if (mdl.P <= kYLds)
    for (int i = threadIdx.x; i < mdl.P * kI; i += blockDim.x) sh.ys[i] = mdl.y[i];
  if (threadIdx.x < kP) {  // col_of: column c works on patch min(c, cpt - 1) % P
    const int cpt = (kP / mdl.P) * mdl.P;
    sh.ypatch[threadIdx.x] = ((threadIdx.x < cpt ? (int)threadIdx.x : cpt - 1) % mdl.P) * kI * (int)sizeof(float);
  }
  __syncthreads();
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


// ---- one pass over the dictionary ---------------------------------------------------------------------------------------
constexpr int kPassG1 = 0;     // residual only:  R = B x - y                                  (head of a trajectory; E(x))
constexpr int kPassFused = 1;  // kick with the residual held, drift, residual at the new x    (one leapfrog step)
constexpr int kPassG2 = 2;     // kick only                                                    (closing half kick; dE/dX)

// R = -y[patch][this wave's 32 pixel rows]
__device__ __forceinline__ void resid_init(const SicModel& mdl, SicShared& sh, int w, int h, int c, int patch, RTile& R) {
  using lds_f32x4 = __attribute__((address_space(3))) const f32x4;
  using glb_f32x4 = __attribute__((address_space(1))) const f32x4;
  const int off = 32 * w + 4 * h;
  if (mdl.P <= kYLds) {
    lds_f32x4* yv = (lds_f32x4*)(unsigned long)(lds_addr(sh.ys) + (unsigned)sh.ypatch[c] + 4u * (unsigned)off);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = yv[2 * g];
#pragma unroll
      for (int k = 0; k < 4; ++k) R.b[0][4 * g + k] = -v[k];
    }
  } else {  // n_patches 5..32: from global memory (counted loads: the waits on the DMA stream only get longer)
    glb_f32x4* yv = (glb_f32x4*)(__attribute__((address_space(1))) const float*)(mdl.y + kI * patch + off);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = yv[2 * g];
#pragma unroll
      for (int k = 0; k < 4; ++k) R.b[0][4 * g + k] = -v[k];
    }
  }
}

// the prior's force of one block: lambda * 2a / (1 + a^2), or lambda * sign(a)
template <bool CAUCHY>
__device__ __forceinline__ void prior_kick(const SicModel& mdl, const f32x16& xb, float scale, f32x16& acc) {
  const float sl = scale * mdl.lambda;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const float a = xb[q];
    // v_rcp_f32: 1 ulp.  Explicit FMAs (the file is compiled without contraction): this is on the critical path of a
    // round -- the waves that run it have just stood at the vector-memory port for their 16 requests
    if (CAUCHY) acc[q] = __builtin_fmaf((sl * 2.0f) * a, __builtin_amdgcn_rcpf(__builtin_fmaf(a, a, 1.0f)), acc[q]);
    else acc[q] += sl * (a > 0.f ? 1.0f : (a < 0.f ? -1.0f : 0.0f));
  }
}

// G2 of an owner's round: acc (a block of V, or of dE/dX) += B^T[block] . (the scaled residual held in pubR)
__device__ __forceinline__ void round_g2(SicShared& sh, const AStream& as, int lane, f32x16& acc) {
  using lds_f32x4 = __attribute__((address_space(3))) const f32x4;
  // the operands of k-step ks + 1 are read before the MFMA of k-step ks; the fences keep hipcc from hoisting all the
  // reads to the top (registers) or sinking them to their uses (an LDS round trip in front of every MFMA)
  const unsigned base = as.img_own;
  // (opaque: as a known constant | lane offset hipcc formed every read's address with a v_or of its own -- the fragment
  // buffers sit beyond the 64 KB an LDS instruction's offset field reaches from zero; from this register it reaches them)
  unsigned rb0 = lds_addr(&sh.pubR[0][0]) + 16u * (unsigned)lane;
  asm volatile("" : "+v"(rb0));
  // kAhead k-steps of operands in flight: during an owner's G2 its SIMD partner has no matrix work, so nothing but the
  // wave's own earlier reads covers the LDS round trip
  constexpr int kAhead = 3;   // (2 and 5 measured: within 2 %)
  f32x4 fa[kAhead], fb[kAhead];
#pragma unroll
  for (int k = 0; k < kAhead; ++k) {
    fa[k] = *(lds_f32x4*)(unsigned long)(base + (unsigned)k * 1024u + as.row[k & 1]);
    fb[k] = *(lds_f32x4*)(unsigned long)(rb0 + (unsigned)k * 1024u);
  }
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const f32x4 a = fa[ks % kAhead], b = fb[ks % kAhead];
    __builtin_amdgcn_sched_barrier(0);
    if (ks + kAhead < 16) {
      fa[ks % kAhead] = *(lds_f32x4*)(unsigned long)(base + (unsigned)(ks + kAhead) * 1024u + as.row[(ks + kAhead) & 1]);
      fb[ks % kAhead] = *(lds_f32x4*)(unsigned long)(rb0 + (unsigned)(ks + kAhead) * 1024u);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ f32x4 frag_scaled(const f32x16& acc, int s, float scale) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (__bf16)(acc[8 * s + j] * scale);
  return __builtin_bit_cast(f32x4, f);
}

// G1 of one round: R += B[this wave's 32 pixel rows, the round's four blocks] . X(those blocks)
template <int RD>
__device__ __forceinline__ void round_g1(SicShared& sh, const AStream& as, int lane, RTile& R) {
  using lds_f32x4 = __attribute__((address_space(3))) const f32x4;
  using lds_i16x4 = __attribute__((address_space(3))) i16x4;
  const unsigned img0 = lds_addr(&sh.img[RD & 1][0][0][0]);
  unsigned xb0 = lds_addr(&sh.pubX[0][0][0]) + 16u * (unsigned)lane;
  asm volatile("" : "+v"(xb0));   // (as rb0 in round_g2)
  // MFMA n = 2 i + s: the fragment A[pixel row][8 coefficients of k-step s of owner i's block] is two transposed
  // 4-coefficient pieces; the operands of MFMA n + 1 are read before MFMA n
  auto ops_of = [&](int n, i16x4& lo, i16x4& hi, f32x4& xf) {
    const unsigned at = img0 + (unsigned)(n >> 1) * 16384u + (unsigned)(n & 1) * 256u;
    lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(unsigned long)(at + as.tr[0]));
    hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_i16x4*)(unsigned long)(at + as.tr[1]));
    xf = *(lds_f32x4*)(unsigned long)(xb0 + (unsigned)n * 1024u);
  };
  // kAheadG1 MFMAs' operands in flight, as round_g2 keeps them: written as "read n + 1, then MFMA n" without a fence between
  // the two, hipcc issued MFMA n FIRST and waited for the reads of n + 1 with lgkmcnt(0) right behind them -- every MFMA
  // of G1 stood behind a whole LDS round trip (the ISA: read, read, read, s_waitcnt lgkmcnt(0), v_mfma, eight times over)
  constexpr int kAheadG1 = 2;
  i16x4 lo[kAheadG1], hi[kAheadG1];
  f32x4 xf[kAheadG1];
#pragma unroll
  for (int k = 0; k < kAheadG1; ++k) ops_of(k, lo[k], hi[k], xf[k]);
#pragma unroll
  for (int n = 0; n < 2 * kG; ++n) {
    const i16x4 l = lo[n % kAheadG1], h = hi[n % kAheadG1];
    const f32x4 x = xf[n % kAheadG1];
    __builtin_amdgcn_sched_barrier(0);
    if (n + kAheadG1 < 2 * kG) ops_of(n + kAheadG1, lo[n % kAheadG1], hi[n % kAheadG1], xf[n % kAheadG1]);
    const auto a8 = __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
    R.b[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, x), R.b[0], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// One round of a pass (RD compile time: the register blocks it touches and the image group are static).  Round RD belongs
// to group RD & 1: each of its four waves owns block RD >> 1 of its coefficient rows.
template <int KIND, bool CAUCHY, int NB, int RD>
__device__ __forceinline__ void pass_round(const SicModel& mdl, SicShared& sh, const AStream& as, int w, int lane,
                                           CTile<NB>& x, CTile<NB>& acc, RTile& R, float scale, float eps, float next_scale) {
  constexpr int T = RD >> 1;
  const bool own = (w >> 2) == (RD & 1);   // wave-uniform
  if constexpr (KIND == kPassFused) SIC_STAMP(RD, 0);
  if (own) {
    wait_vm<0>();                // this wave's block image has landed (issued a round ago)
    if constexpr (KIND != kPassG1) {
      round_g2(sh, as, lane, acc.b[T]);
    }
    if constexpr (KIND == kPassFused) SIC_STAMP(RD, 1);
    if constexpr (KIND == kPassG2) {
      issue_block<NB, T + 1>(as);   // nobody else reads this image in a kick-only pass
    } else {
      if constexpr (KIND == kPassFused) x.b[T] = x.b[T] + eps * acc.b[T];   // drift (acc is V)
      sh.pubX[w & 3][0][lane] = frag_scaled(x.b[T], 0, 1.0f);
      sh.pubX[w & 3][1][lane] = frag_scaled(x.b[T], 1, 1.0f);
      // the tail of the partner's request burst (it owned round RD - 1 and asked for block (RD - 1) / 2 + 1 after that
      // round's barrier B); landed before barrier B of this round, after which the partner's G2 reads the image
      if constexpr (KIND == kPassFused && RD >= 1 && kOwnPieces < 16)
        issue_partner<NB, ((RD - 1) >> 1) + 1, kOwnPieces, 16, (RD & 1) ? -1 : 1>(as);
    }
  } else if constexpr (KIND != kPassG1 && RD + 1 < 2 * NB) {
    // beside the owners' matrix work: the prior's force of the block this wave owns in the NEXT round
    prior_kick<CAUCHY>(mdl, x.b[(RD + 1) >> 1], scale, acc.b[(RD + 1) >> 1]);
  } else if constexpr (KIND != kPassG2 && RD + 1 == 2 * NB) {
    // the pass's last round: group 0 has no next block in this pass -- the prior's force of its block 0 for the NEXT kick
    // (x's block 0 is final since round 0, and so is this pass's kick of it); at the head of that pass it would stand
    // alone, 400 cycles with the other group waiting at the barrier.  next_scale == 0: no kick follows
    if (next_scale != 0.f) prior_kick<CAUCHY>(mdl, x.b[0], next_scale, acc.b[0]);
  }
  if constexpr (KIND == kPassG2) return;
  if constexpr (KIND == kPassFused) SIC_STAMP(RD, 2);
  __syncthreads();               // barrier A: the round's four images have landed, their X blocks are published
  if constexpr (KIND == kPassFused) SIC_STAMP(RD, 3);
  round_g1<RD>(sh, as, lane, R);
  if constexpr (KIND == kPassFused) SIC_STAMP(RD, 4);
  if constexpr (KIND == kPassFused && RD >= 1 && kOwnPieces < 16) {
    if (own) wait_vm<0>();                      // (the partner's pieces: requested at least a whole G1 ago)
  }
  __syncthreads();               // barrier B: the round's images and X fragments have been read by every wave
  if constexpr (KIND == kPassFused) SIC_STAMP(RD, 5);
  if (own) {   // the owner's next block (of the next pass after the last one): lands during the other group's round
    if constexpr (KIND == kPassFused && RD + 1 < 2 * NB) issue_block<NB, T + 1, 0, kOwnPieces>(as);
    else issue_block<NB, T + 1>(as);
  }
  if constexpr (KIND == kPassFused) SIC_STAMP(RD, 6);
}

template <int KIND, bool CAUCHY, int NB, int RD>
struct Rounds {
  static __device__ __forceinline__ void run(const SicModel& mdl, SicShared& sh, const AStream& as, int w, int lane,
                                             CTile<NB>& x, CTile<NB>& acc, RTile& R, float scale, float eps, float next_scale) {
    pass_round<KIND, CAUCHY, NB, RD>(mdl, sh, as, w, lane, x, acc, R, scale, eps, next_scale);
    if constexpr (RD + 1 < 2 * NB) Rounds<KIND, CAUCHY, NB, RD + 1>::run(mdl, sh, as, w, lane, x, acc, R, scale, eps, next_scale);
  }
};

// kPassG1:    R = B x - y.
// kPassFused: acc (= V) += scale * dE/dX at the x whose residual is R;  x += eps * V;  R = B x - y at the new x.
// kPassG2:    acc += scale * dE/dX at the x whose residual is R (x, R unchanged).
// HEAD_PRIOR: group 0's block-0 prior force is added at the head of this (kick) pass; false when the pass before has
// done it in its last round (its `next_scale` = this pass's scale; 0 = no kick pass follows).
template <int KIND, bool CAUCHY, int NB, bool HEAD_PRIOR = true>
__device__ __forceinline__ void sic_pass(const SicModel& mdl, SicShared& sh, const AStream& as, int w, int c, int h, int lane,
                                         int patch, CTile<NB>& x, CTile<NB>& acc, RTile& R, float scale, float eps,
                                         float next_scale = 0.f) {
  if constexpr (KIND != kPassG1) {
    // the residual, scaled by the step and by 1 / n_patches (d/da_p of the MEAN over patches), as the B operand of G2:
    // every wave publishes its two k-steps (the last readers of pubR were the G2 rounds of the previous kick pass, which
    // every wave has left through a barrier since)
    const float sc = scale * mdl.invP;
    sh.pubR[2 * w + 0][lane] = frag_scaled(R.b[0], 0, sc);
    sh.pubR[2 * w + 1][lane] = frag_scaled(R.b[0], 1, sc);
    if (HEAD_PRIOR && (w >> 2) == 0) prior_kick<CAUCHY>(mdl, x.b[0], scale, acc.b[0]);   // group 0 owns round 0: nobody to do it beside
    __syncthreads();
  }
  if constexpr (KIND != kPassG2) resid_init(mdl, sh, w, h, c, patch, R);
  Rounds<KIND, CAUCHY, NB, 0>::run(mdl, sh, as, w, lane, x, acc, R, scale, eps, next_scale);
#ifdef SIC_STAMPS
  if constexpr (KIND == kPassFused) {   // (the last pass run wins)
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 256) (&g_sic_stamp[0][0][0])[threadIdx.x] = (&sh.stamp[0][0][0])[threadIdx.x];
  }
#endif
  if constexpr (KIND == kPassG2) __syncthreads();   // every wave's G2 reads of pubR are done before the next pass rewrites it
}

__device__ __forceinline__ float half_swap_sum(float s) {
  const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_int(s), __float_as_int(s), false, false);
  return __int_as_float(sw[0]) + __int_as_float(sw[1]);
}

// sum of a per-column value over the columns of the caller's particle (n_patches of them, consecutive)
template <class SH>
__device__ __forceinline__ float group_total(SH& sh, int w, int c, int h, int P, int g0, float col_tot) {
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
__device__ __forceinline__ float sic_energy(const SicModel& mdl, SicShared& sh, int w, int c, int h, const Col& col,
                                            const RTile& R, const CTile<NB>& x) {
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += 0.5f * R.b[0][q] * R.b[0][q];
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
  for (int k = 0; k < kW; ++k) tot += sh.red[0][k][c];
  __syncthreads();
  return group_total(sh, w, c, h, mdl.P, col.g0, tot);
}

// sum(v^2) / 2 per PARTICLE
template <int NB, class SH>
__device__ __forceinline__ float sic_kinetic(SH& sh, int w, int c, int h, int P, const Col& col, const CTile<NB>& v) {
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
  for (int k = 0; k < kW; ++k) tot += sh.red[1][k][c];
  __syncthreads();
  return group_total(sh, w, c, h, P, col.g0, tot) / 2.0f;
}

// what storing a tile in the state's type does to its values (the energies reported are those of the STORED state)
template <typename ST, int NB>
__device__ __forceinline__ void round_to_state(CTile<NB>& t) {
  if constexpr (sizeof(ST) == 2) {
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) t.b[b][q] = (float)(__bf16)t.b[b][q];
  }
}

// L leapfrog steps with the half kicks between drifts merged (bf16 operands make the reference's
// separate roundings meaningless).  Returns E(x_new) of the particle; x, v updated in place; R = residual at x_new.
template <bool CAUCHY, int NB, typename ST>
__device__ __forceinline__ float sic_trajectory(const SicModel& mdl, SicShared& sh, const AStream& as, int w, int c, int h,
                                                int lane, const Col& col, CTile<NB>& x, CTile<NB>& v, RTile& R, int L,
                                                float eps, float chalf) {
  sic_pass<kPassG1, CAUCHY, NB>(mdl, sh, as, w, c, h, lane, col.patch, x, v, R, 0.f, 0.f, L > 0 ? chalf : 0.f);
  if (L > 0) {
    // the step scale is wave-uniform: both values sit in scalar registers, the select is an integer s_cselect
    const int c_half = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, chalf));
    const int c_full = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, 2.0f * chalf));
#pragma unroll 1
    for (int s = 1; s <= L; ++s) {
      asm volatile("; MJHMC_LEAPFROG_STEP_BEGIN (tools/check_isa.sh)");
      // kick (half the first time, two merged halves afterwards), drift, residual at the new position
      sic_pass<kPassFused, CAUCHY, NB, false>(mdl, sh, as, w, c, h, lane, col.patch, x, v, R,
                                               __builtin_bit_cast(float, s == 1 ? c_half : c_full), eps,
                                               __builtin_bit_cast(float, s == L ? c_half : c_full));
      asm volatile("; MJHMC_LEAPFROG_STEP_END");
    }
    sic_pass<kPassG2, CAUCHY, NB, false>(mdl, sh, as, w, c, h, lane, col.patch, x, v, R, chalf, 0.f);   // closing half kick
  }
  // a bf16 state: the successor position is stored in bf16 (and G1 already saw bf16(x)): evaluate the prior on
  // what will be stored, so EX is the energy of the stored state
  round_to_state<ST>(x);
  return sic_energy<CAUCHY, NB>(mdl, sh, w, c, h, col, R, x);
}

// v += mix * (standard normals of this lane's dims of particle `pid`; dims patch * n_coeffs + ...), group by group: one
// Box-Muller quadruple's temporaries at a time
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

template <int NB>
__device__ __forceinline__ void ctile_zero(CTile<NB>& t) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int q = 0; q < 16; ++q) t.b[b][q] = 0.f;
}

// ---------------------------------------------------------------------------------------------------
template <bool CAUCHY, int NB, typename ST>
__global__ __launch_bounds__(512, 2) void sic_eval_kernel(const SicEvalArgsT<ST> a, const SicModel mdl) {
  __shared__ SicShared sh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  stage_patches(mdl, sh);
  const AStream as = astream_open<NB>(mdl, sh, w, lane);
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const Col col = col_of(tile, c, mdl.P, a.N, Identity{});
    CTile<NB> x, g;
    RTile R;
    ctile_load(a.X, col.q, w, h, x);
    sic_pass<kPassG1, CAUCHY, NB>(mdl, sh, as, w, c, h, lane, col.patch, x, g, R, 0.f, 0.f);
    if (a.G) {
      ctile_zero(g);
      sic_pass<kPassG2, CAUCHY, NB>(mdl, sh, as, w, c, h, lane, col.patch, x, g, R, 1.0f, 0.f);
      if (col.alive) ctile_store_f32(a.G, col.q, w, h, g);
    }
    const float ex = sic_energy<CAUCHY, NB>(mdl, sh, w, c, h, col, R, x);
    if (a.E && w == 0 && h == 0 && col.leader) a.E[col.part] = ex;
    if (a.EV) {
      CTile<NB> v;
      if (a.V_gen) {
        ctile_zero(v);
        sic_add_normals(a.key, (uint32_t)(a.first_pid + col.part), col.patch, w, h, 1.0f, v);
        round_to_state<ST>(v);  // EV matches the stored momentum
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

// The list of the FIRST iteration of a call: the particles whose inverse-L proposal must be integrated (cold cache and
// no H(L proposal) handed on by an F move; later lists: append_cold in the jump and fix kernels; dense_pot.hip has the
// story of the F-movers, of the inverse-L tiles as items of the jump launch and of the fix kernel)
__global__ void sic_cold_list_kernel(const float* __restrict__ Hflf_in, const float* __restrict__ Hspec_in, int64_t N,
                                     int* __restrict__ list, int* __restrict__ count, const Control* ctl) {
  if (ctl->failed) return;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float hc = p < N ? Hflf_in[p] : 0.f, hs = p < N ? Hspec_in[p] : 0.f;
  append_cold(list, count, (p < N) && !(hc == hc) && !(hs == hs), p);
}

struct FromList {
  const int* list;
  __device__ int64_t operator()(int64_t s) const { return list[s]; }
};

// what the kernels that only FINISH a move need of SicShared (sic_fix_kernel)
struct SicFinishShared {
  float red[2][kW][kP];
  float colsum[kP];
  int move[kP];
  unsigned tally[4];
};

// The successor's rows once the moves of a tile's columns stand in sh.move.  FIX = false (jump kernel): x, v hold the
// end point of L.  FIX = true (sic_fix_kernel): columns that keep the end point are finished already.
template <bool REPLAY, int MODE, int NB, bool FIX, class SH, typename ST>
__device__ __forceinline__ void sic_finish(const SicJumpArgsT<ST>& a, SH& sh, int P, const Col& col, int w, int c, int h,
                                           CTile<NB>& x, CTile<NB>& v) {
  const int64_t p = col.part;
  const int mv = sh.move[c];
  const int k = mv & 3;
  bool refresh;  // this column's momentum is redrawn (HMCState.R)
  bool touch = col.alive;
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
    if constexpr (FIX) touch = touch && k != 0;
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
      CTile<NB> z;   // the recorded normals, stored like the state
      ctile_load(a.noise, col.q, w, h, z);
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < 16; ++q) v.b[b][q] += z.b[b][q] * a.r_mix;
    } else {
      sic_add_normals(a.key, (uint32_t)(a.first_pid + p), col.patch, w, h, a.r_mix, v);
    }
    round_to_state<ST>(v);
  }
  if (tile_refreshes) {
    const float evr = sic_kinetic(sh, w, c, h, P, col, v);
    if (refresh && w == 0 && h == 0 && col.leader) a.EV_out[p] = evr;
  }
  if (touch) {
    ctile_store(a.X_out, col.q, w, h, x);
    ctile_store(a.V_out, col.q, w, h, v);
  }
}

// MODE = kModeMJHMC (markov_jump_hmc.py:355-415), kModeCT (:251-290) or kModeControl (:116-148, the comparison arm of
// the reference's sparse-coding experiments, search/control_sp_img/control_objective.py:10)
template <bool CAUCHY, bool REPLAY, int MODE, int NB, typename ST>
__global__ __launch_bounds__(512, 2) void sic_jump_kernel(const SicJumpArgsT<ST> a, const SicModel mdl) {
  __shared__ SicShared sh;
  if (a.ctl->failed) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  // MJHMC: the inverse-L tiles of this iteration's list are the first items of the launch
  const int ncold = MODE == kModeMJHMC ? *a.cold_count : 0;
  const int ppt = kP / mdl.P;
  const int64_t nft = (ncold + ppt - 1) / ppt;
  if ((int64_t)blockIdx.x >= nft + a.ntiles) return;
  if (MODE == kModeMJHMC && blockIdx.x == 0 && threadIdx.x == 0) {
    *a.zero_count = 0;   // the list two iterations back is consumed: its counter is free for the next iteration's appends
    if (ncold) atomicAdd(&a.stats[3], (unsigned long long)ncold << 32);   // integrated here: the high half of the cold tally
  }
  unsigned n0 = 0, n1 = 0, n2 = 0, n3 = 0;  // tallies (meaning per mode: fill_iter_stats in api.hip)
  bool any_bad = false;
  if (threadIdx.x < 4) sh.tally[threadIdx.x] = 0;
  stage_patches(mdl, sh);
  const AStream as = astream_open<NB>(mdl, sh, w, lane);
  for (int64_t item = blockIdx.x; item < nft + a.ntiles; item += gridDim.x) {
    const bool inverse = item < nft;   // (uniform over the workgroup)
    const Col col = inverse ? col_of(item, c, mdl.P, (int64_t)ncold, FromList{a.cold_list})   // the last tile repeats an entry
                            : col_of(item - nft, c, mdl.P, a.N, Identity{});
    const int64_t p = col.part;
    CTile<NB> x, v;
    RTile R;
    ctile_load(a.X_in, col.q, w, h, x);
    ctile_load(a.V_in, col.q, w, h, v);
    if (inverse) {
#pragma unroll
      for (int b = 0; b < NB; ++b) v.b[b] = -v.b[b];
    }
    const float EXL = sic_trajectory<CAUCHY, NB, ST>(mdl, sh, as, w, c, h, lane, col, x, v, R, a.L, a.eps, a.chalf);
    round_to_state<ST>(v);  // the successor state is stored in bf16: report the kinetic energy of what is stored
    const float EVL = sic_kinetic(sh, w, c, h, mdl.P, col, v);
    const float HL = EXL + EVL;
    if (inverse) {
      if (w == 0 && h == 0 && col.leader) a.Hwork[p] = HL;
      __syncthreads();
      continue;
    }
    if (w == 0 && h == 0) {  // every column of a particle reaches the same decision; its leader reports it
      const uint32_t pid = (uint32_t)(a.first_pid + p);
      const float EX0 = a.EX_in[p], EV0 = a.EV_in[p];
      const float H0 = EX0 + EV0;
      // H of the inverse-L proposal: cached; or the L proposal of the iteration in which the particle flipped; or being
      // integrated by an inverse-L item of this very launch -- then the particle is left pending for sic_fix_kernel
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
        if constexpr (MODE == kModeMJHMC) k = dense_decide<REPLAY>(H0, HL, Hflf, a.p_r, pid, p, a.N, a.rexp, a.key, best, bad);
        else if constexpr (MODE == kModeCT) k = dense_decide_ct<REPLAY>(H0, HL, a.p_r, pid, p, a.N, a.rexp, a.key, best, bad);
        else k = dense_control<REPLAY>(H0, HL, a.p_r, a.p_flip, pid, p, a.N, a.runif, a.key, gate);
        // every move but L clears the cache; of those only the R-movers need their inverse-L proposal integrated
        if constexpr (MODE == kModeMJHMC) append_cold(a.next_list, a.next_count, col.leader && k == 2, p);
      }
      sh.move[c] = k | (gate ? 4 : 0);
      if (col.leader) {
        if (!pending) {
          any_bad |= bad;
          a.dwell[p] = best;
          a.dwell_ring[p] = best;
          a.trans[p] = (uint8_t)k;
        }
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
        const bool took_L = MODE == kModeControl ? (k & 1) : (k == 0);
        a.EX_out[p] = took_L ? EXL : EX0;
        a.EV_out[p] = took_L ? EVL : EV0;
        if (!pending) {
          a.Hflf_out[p] = (MODE == kModeMJHMC && k == 0) ? H0 : __builtin_nanf("");
          if constexpr (MODE == kModeMJHMC) a.Hspec_out[p] = (k == 1) ? HL : __builtin_nanf("");
        }
      }
    }
    __syncthreads();
    sic_finish<REPLAY, MODE, NB, false>(a, sh, mdl.P, col, w, c, h, x, v);
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

// The particles the jump kernel left pending (this iteration's list): decide them now that both of their trajectories
// are done, and where the move is not L put the pre-move position back and flip / redraw the momentum (pot_fix_kernel's twin)
template <bool REPLAY, int NB, typename ST>
__global__ __launch_bounds__(512) void sic_fix_kernel(const SicJumpArgsT<ST> a, int P) {
  __shared__ SicFinishShared sh;
  if (a.ctl->failed) return;
  const int ncold = *a.cold_count;
  const int ppt = kP / P;
  if ((int64_t)blockIdx.x * ppt >= ncold) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  unsigned n0 = 0, n1 = 0, n2 = 0;
  bool any_bad = false;
  if (threadIdx.x < 4) sh.tally[threadIdx.x] = 0;
  for (int64_t tile = blockIdx.x; tile * ppt < ncold; tile += gridDim.x) {
    const Col col = col_of(tile, c, P, (int64_t)ncold, FromList{a.cold_list});   // idle columns: alive == false
    const int64_t p = col.part;
    if (w == 0 && h == 0) {
      const uint32_t pid = (uint32_t)(a.first_pid + p);
      const float EX0 = a.EX_in[p], EV0 = a.EV_in[p];
      const float H0 = EX0 + EV0;
      const float HL = a.EX_out[p] + a.EV_out[p];
      double best = 0.0;
      bool bad = false;
      const int k = dense_decide<REPLAY>(H0, HL, a.Hwork[p], a.p_r, pid, p, a.N, a.rexp, a.key, best, bad);
      sh.move[c] = col.alive ? k : 0;
      __builtin_amdgcn_wave_barrier();   // (every column has read EX_out / EV_out before a leader overwrites them)
      append_cold(a.next_list, a.next_count, col.leader && k == 2, p);
      if (col.leader) {
        any_bad |= bad;
        a.dwell[p] = best;
        a.dwell_ring[p] = best;
        a.trans[p] = (uint8_t)k;
        n0 += (k == 0);
        n1 += (k == 1);
        n2 += (k == 2);
        if (k != 0) {
          a.EX_out[p] = EX0;
          a.EV_out[p] = EV0;   // (an R-mover's: filled in by sic_finish)
        }
        a.Hflf_out[p] = k == 0 ? H0 : __builtin_nanf("");
        a.Hspec_out[p] = k == 1 ? HL : __builtin_nanf("");
      }
    }
    __syncthreads();
    CTile<NB> x, v;
    ctile_zero(v);
    sic_finish<REPLAY, kModeMJHMC, NB, true>(a, sh, P, col, w, c, h, x, v);
    __syncthreads();
  }
  if (any_bad) {
    a.ctl->failed = 1;
    a.ctl->failed_iter = a.iter;
  }
  if (n0) atomicAdd(&sh.tally[0], n0);
  if (n1) atomicAdd(&sh.tally[1], n1);
  if (n2) atomicAdd(&sh.tally[2], n2);
  __syncthreads();
  if (threadIdx.x < 3 && sh.tally[threadIdx.x]) atomicAdd(&a.stats[threadIdx.x], (unsigned long long)sh.tally[threadIdx.x]);
}

// HMCState.leapfrog / HMCState.L on caller-supplied states (hmc_state.py:86-100)
template <bool CAUCHY, int NB, typename ST>
__global__ __launch_bounds__(512, 2) void sic_leap_kernel(const SicLeapArgsT<ST> a, const SicModel mdl) {
  __shared__ SicShared sh;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = lane & 31, h = lane >> 5;
  stage_patches(mdl, sh);
  const AStream as = astream_open<NB>(mdl, sh, w, lane);
  for (int64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const Col col = col_of(tile, c, mdl.P, a.N, Identity{});
    CTile<NB> x, v;
    RTile R;
    ctile_load(a.X, col.q, w, h, x);
    ctile_load(a.V, col.q, w, h, v);
    const float ex = sic_trajectory<CAUCHY, NB, ST>(mdl, sh, as, w, c, h, lane, col, x, v, R, a.L, a.eps, a.chalf);
    round_to_state<ST>(v);
    const float ev = sic_kinetic(sh, w, c, h, mdl.P, col, v);
    if (col.alive) {
      ctile_store(a.X_out, col.q, w, h, x);
      ctile_store(a.V_out, col.q, w, h, v);
    }
    if (w == 0 && h == 0 && col.leader) {
      if (a.EX) a.EX[col.part] = ex;
      if (a.EV) a.EV[col.part] = ev;
    }
    if (a.G) {  // dE/dX of the stored end point, into the registers the momentum has just left
      sic_pass<kPassG1, CAUCHY, NB>(mdl, sh, as, w, c, h, lane, col.patch, x, v, R, 0.f, 0.f);
      ctile_zero(v);
      sic_pass<kPassG2, CAUCHY, NB>(mdl, sh, as, w, c, h, lane, col.patch, x, v, R, 1.0f, 0.f);
      if (col.alive) ctile_store_f32(a.G, col.q, w, h, v);
    }
    __syncthreads();
  }
  astream_close();
}

#ifdef SIC_STAMPS
}  // namespace mjhmc
extern "C" int mjhmc_sic_stamps(unsigned* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mjhmc::g_sic_stamp), sizeof(mjhmc::g_sic_stamp));
}
namespace mjhmc {
#endif

static int sic_cus() {
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return std::max(1, cus);
}

template <bool CAUCHY, int MODE, int NB, typename ST>
static void sic_launch_mode(const SicJumpArgsT<ST>& a, const SicModel& mdl, unsigned grid, hipStream_t st) {
  const bool replay = MODE == kModeControl ? (a.runif && a.noise) : (a.rexp && a.noise);
  if (replay) hipLaunchKernelGGL((sic_jump_kernel<CAUCHY, true, MODE, NB, ST>), dim3(grid), dim3(512), 0, st, a, mdl);
  else hipLaunchKernelGGL((sic_jump_kernel<CAUCHY, false, MODE, NB, ST>), dim3(grid), dim3(512), 0, st, a, mdl);
}

template <bool CAUCHY, int NB, typename ST>
static void sic_launch_jump_t(const SicJumpArgsT<ST>& a, const SicModel& mdl, hipStream_t st) {
  const int cus = sic_cus();
  if (a.mode == kModeMJHMC) {  // only MJHMC has the inverse-L proposal and its cache
    if (a.iter == 0 || a.rescan) {  // first iteration of a call: the three counters cleared (they are adjacent), the list from a scan
      (void)hipMemsetAsync(std::min(a.cold_count, std::min(a.next_count, a.zero_count)), 0, 3 * sizeof(int), st);
      hipLaunchKernelGGL(sic_cold_list_kernel, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, st, a.Hflf_in, a.Hspec_in,
                         a.N, a.cold_list, a.cold_count, (const Control*)a.ctl);
    }
    // forward tiles + at most as many inverse-L tiles (workgroups without an item leave at once)
    const unsigned grid = (unsigned)std::min<int64_t>(2 * a.ntiles, cus);
    sic_launch_mode<CAUCHY, kModeMJHMC, NB, ST>(a, mdl, grid, st);
    const unsigned fgrid = (unsigned)std::min<int64_t>(a.ntiles, 4 * cus);
    if (a.rexp && a.noise) hipLaunchKernelGGL((sic_fix_kernel<true, NB, ST>), dim3(fgrid), dim3(512), 0, st, a, mdl.P);
    else hipLaunchKernelGGL((sic_fix_kernel<false, NB, ST>), dim3(fgrid), dim3(512), 0, st, a, mdl.P);
  } else {
    const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, cus);
    if (a.mode == kModeCT) sic_launch_mode<CAUCHY, kModeCT, NB, ST>(a, mdl, grid, st);
    else sic_launch_mode<CAUCHY, kModeControl, NB, ST>(a, mdl, grid, st);
  }
}

// the prior (Cauchy / Laplace) and the dictionary width (1024 / 512 atoms) select the instantiation
#define SIC_DISPATCH(CALL)                                   \
  do {                                                       \
    if (mdl.nc == 1024) {                                    \
      if (mdl.cauchy) CALL(true, 4); else CALL(false, 4);    \
    } else {                                                 \
      if (mdl.cauchy) CALL(true, 2); else CALL(false, 2);    \
    }                                                        \
  } while (0)

template <typename ST>
static void sic_launch_jump_st(const SicJumpArgsT<ST>& a, const SicModel& mdl, hipStream_t st) {
#define SIC_JUMP(C, NBV) sic_launch_jump_t<C, NBV, ST>(a, mdl, st)
  SIC_DISPATCH(SIC_JUMP);
}
template <typename ST>
static void sic_launch_eval_st(const SicEvalArgsT<ST>& a, const SicModel& mdl, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, sic_cus());
#define SIC_EVAL(C, NBV) hipLaunchKernelGGL((sic_eval_kernel<C, NBV, ST>), dim3(grid), dim3(512), 0, st, a, mdl)
  SIC_DISPATCH(SIC_EVAL);
}
template <typename ST>
static void sic_launch_leap_st(const SicLeapArgsT<ST>& a, const SicModel& mdl, hipStream_t st) {
  const unsigned grid = (unsigned)std::min<int64_t>(a.ntiles, sic_cus());
#define SIC_LEAP(C, NBV) hipLaunchKernelGGL((sic_leap_kernel<C, NBV, ST>), dim3(grid), dim3(512), 0, st, a, mdl)
  SIC_DISPATCH(SIC_LEAP);
}

void sic_launch_jump(const SicJumpArgsT<__bf16>& a, const SicModel& mdl, hipStream_t st) { sic_launch_jump_st(a, mdl, st); }
void sic_launch_eval(const SicEvalArgsT<__bf16>& a, const SicModel& mdl, hipStream_t st) { sic_launch_eval_st(a, mdl, st); }
void sic_launch_leap(const SicLeapArgsT<__bf16>& a, const SicModel& mdl, hipStream_t st) { sic_launch_leap_st(a, mdl, st); }
void sic_launch_jump(const SicJumpArgsT<float>& a, const SicModel& mdl, hipStream_t st) { sic_launch_jump_st(a, mdl, st); }
void sic_launch_eval(const SicEvalArgsT<float>& a, const SicModel& mdl, hipStream_t st) { sic_launch_eval_st(a, mdl, st); }
void sic_launch_leap(const SicLeapArgsT<float>& a, const SicModel& mdl, hipStream_t st) { sic_launch_leap_st(a, mdl, st); }

}  // namespace mjhmc
