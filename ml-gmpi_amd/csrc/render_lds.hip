// render_lds.hip -- GMPI_VARIANT_LDS (placeholder until the LDS-staged kernel lands)
#include "gmpi_device.hpp"
namespace gmpi {
bool lds_variant_supports(const KParams&, int) { return false; }
int lds_variant_query(int) { return 0; }
hipError_t launch_lds(const KParams&, int, hipStream_t) { return hipErrorNotSupported; }
}  // namespace gmpi
