// Instantiations of the wave-per-trajectory core kernel (d > 64), strict fp64 math.
#include "mfg_core.h"
namespace mfg {
int launch_core_large_f64(const CoreArgs& a, bool sample, bool td, int num_cus, hipStream_t st) {
  return launch_core_large_impl<false>(a, sample, td, num_cus, st);
}
}  // namespace mfg
