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
