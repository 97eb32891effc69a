// Funnel functors (Funnel, mjhmc/misc/tf_distributions.py:142-177): Neal's density and the
// energy exactly as coded.
#include "elementwise.hpp"
namespace mjhmc {
static inline FunnelNealF<double> make_fn64(const EnergyParams& ep) {
  return FunnelNealF<double>{1.0 / (ep.p[0] * ep.p[0]), 0.5 * (ep.ndims - 1)};
}
static inline FunnelNealF<float> make_fn32(const EnergyParams& ep) {
  return FunnelNealF<float>{(float)(1.0 / (ep.p[0] * ep.p[0])), (float)(0.5 * (ep.ndims - 1))};
}
static inline FunnelRefF<double> make_fr64(const EnergyParams& ep) {
  return FunnelRefF<double>{1.0 / (ep.p[0] * ep.p[0]), (double)(ep.ndims - 1)};
}
static inline FunnelRefF<float> make_fr32(const EnergyParams& ep) {
  return FunnelRefF<float>{(float)(1.0 / (ep.p[0] * ep.p[0])), (float)(ep.ndims - 1)};
}
MJHMC_DEFINE_ENERGY_LAUNCHERS(funnel_neal, make_fn64, make_fn32)
MJHMC_DEFINE_ENERGY_LAUNCHERS(funnel_ref, make_fr64, make_fr32)
}  // namespace mjhmc

#ifdef ROWS_STAMPS
// timing build (tools/rows_stamps.sh): the cycle stamps of timing_variants.hpp
extern "C" int mjhmc_rows_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mjhmc::g_rows_stamp), sizeof(mjhmc::g_rows_stamp));
}
#endif
