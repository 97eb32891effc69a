// MMGaussF instantiations (MultimodalGaussian, mjhmc/misc/distributions.py:314-335).
#include "elementwise.hpp"
namespace mjhmc {
static inline MMGaussF<double> make_mm64(const EnergyParams& ep) { return MMGaussF<double>{2.0 * ep.p[0]}; }
static inline MMGaussF<float> make_mm32(const EnergyParams& ep) { return MMGaussF<float>{(float)(2.0 * ep.p[0])}; }
MJHMC_DEFINE_ENERGY_LAUNCHERS(mm, make_mm64, make_mm32)
}  // namespace mjhmc
