// IsoGaussF instantiations (TestGaussian, mjhmc/misc/distributions.py:348-362).
#include "elementwise.hpp"
namespace mjhmc {
static inline IsoGaussF<double> make_iso64(const EnergyParams& ep) {
  const double s = ep.p[0];
  return IsoGaussF<double>{1.0 / (s * s), 1.0 / (2.0 * (s * s))};
}
static inline IsoGaussF<float> make_iso32(const EnergyParams& ep) {
  const double s = ep.p[0];
  return IsoGaussF<float>{(float)(1.0 / (s * s)), (float)(1.0 / (2.0 * (s * s)))};
}
MJHMC_DEFINE_ENERGY_LAUNCHERS(iso, make_iso64, make_iso32)
}  // namespace mjhmc
