// DiagGaussF instantiations (Gaussian, mjhmc/misc/distributions.py:256-273).
#include "elementwise.hpp"
namespace mjhmc {
static inline DiagGaussF<double> make_diag64(const EnergyParams& ep) {
  return DiagGaussF<double>{(const double*)ep.dev_f64};
}
static inline DiagGaussF<float> make_diag32(const EnergyParams& ep) {
  return DiagGaussF<float>{(const float*)ep.dev_f32};
}
MJHMC_DEFINE_ENERGY_LAUNCHERS(diag, make_diag64, make_diag32)
}  // namespace mjhmc
