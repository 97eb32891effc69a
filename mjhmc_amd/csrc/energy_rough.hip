// RoughWellF instantiations (RoughWell, mjhmc/misc/distributions.py:283-304).
#include "elementwise.hpp"
namespace mjhmc {
static inline RoughWellF<double> make_rough64(const EnergyParams& ep) {
  const double s1 = ep.p[0], s2 = ep.p[1];
  return RoughWellF<double>{s1 * s1, 2.0 * (s1 * s1), s2};
}
static inline RoughWellF<float> make_rough32(const EnergyParams& ep) {
  const double s1 = ep.p[0], s2 = ep.p[1];
  return RoughWellF<float>{(float)(s1 * s1), (float)(2.0 * (s1 * s1)), (float)s2};
}
MJHMC_DEFINE_ENERGY_LAUNCHERS(rough, make_rough64, make_rough32)
}  // namespace mjhmc
