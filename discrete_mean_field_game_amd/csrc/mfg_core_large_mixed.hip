// Instantiations of the wave-per-trajectory core kernel (d > 64), mixed precision.
#include "mfg_core.h"
namespace mfg {
int launch_core_large_mixed(const CoreArgs& a, bool sample, bool td, int num_cus, hipStream_t st) {
  return launch_core_large_impl<true>(a, sample, td, num_cus, st);
}
}  // namespace mfg
