// Compile-time hooks of the TIMING builds (tools/pot_stamps.sh, tools/run_sic_variants.sh).  The product build defines
// none of the switches, and every macro here expands to nothing: the shipped kernel text has no timing branches.
#pragma once

// -DPOT_STAMPS: cycle stamps of workgroup 0's waves around the parts of a ProductOfT gradient evaluation, [wave][slot][part]
// (slot: `stamp_slot` in scope -- the leapfrog step in the float64-state kernel, 0 elsewhere; the last writer wins).
// Six to eight stamps per ~150 000-cycle gradient: they do not disturb what they time.
#ifdef POT_STAMPS
namespace mjhmc {
static __device__ unsigned long long g_pot_stamp[4][8][8];
}
#define POT_STAMP(I)                                                                                          \
  do {                                                                                                        \
    if (blockIdx.x == 0 && lane == 0) g_pot_stamp[w][stamp_slot & 7][I] = __builtin_readcyclecounter();      \
  } while (0)
#else
#define POT_STAMP(I) do { } while (0)
#endif

// -DSIC_STAMPS=<k> (tools/sic_variants.sh <k>): cycle stamps (s_memtime = core clock) of four of workgroup 0's waves at
// position k of every round of a SparseImageCode leapfrog step's pass; tools/sic_leap_time.py reads them,
// tools/sic_stamps_merge.py joins the builds.  ONE position per build: a stamp costs ~150 cycles (s_memtime, its wait,
// the LDS write), seven per round distort what they measure -- plus the entry into round 0, on which the builds are
// aligned.  Kept in LDS (SicShared::stamp) during the pass: a global store would queue behind the dictionary requests it
// is supposed to time.  (The builds that switched PARTS of a round off -- round 3's SICV 1..12, 70..86 -- are gone from
// the source: what they measured is recorded in DESIGN.md section 3.5.)
#ifdef SIC_STAMPS
namespace mjhmc {
static __device__ unsigned g_sic_stamp[4][8][8];
}
#define SIC_STAMP(RD, I)                                                                                          \
  do {                                                                                                            \
    if (((I) == (SIC_STAMPS) || ((I) == 0 && (RD) == 0)) && blockIdx.x == 0 && lane == 0 && (w & 2) == 0)         \
      sh.stamp[(w & 1) | ((w >> 2) << 1)][RD][I] = (unsigned)__builtin_readcyclecounter();                        \
  } while (0)
#else
#define SIC_STAMP(RD, I) do { } while (0)
#endif

// -DROWS_STAMPS (tools/rows_stamps.sh): cycle stamps of workgroup 0's four waves at the phase boundaries of the relay kernel's
// sampling iterations (elementwise.hpp: mjhmc_fused_rows_relay_kernel), first workgroup tile only, [wave][iteration][point].
// Twelve stamps per ~25 000-cycle iteration; tools/rows_stamps.py reads them (mjhmc_rows_stamps in energy_funnel.hip).
#ifdef ROWS_STAMPS
namespace mjhmc {
static __device__ unsigned long long g_rows_stamp[4][64][16];
}
#define ROWS_STAMP(I)                                                                                                    \
  do {                                                                                                                   \
    if (blockIdx.x == 0 && stamp_on && (threadIdx.x & 63) == 0) g_rows_stamp[threadIdx.x >> 6][stamp_it & 63][I] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define ROWS_STAMP(I) do { } while (0)
#endif
