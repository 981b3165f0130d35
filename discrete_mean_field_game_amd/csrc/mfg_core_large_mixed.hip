// Instantiations of the wave-per-trajectory core kernel (d > 64), mixed precision: every R but 2 and 4, which live in
// mfg_core_large_mixed_ilp.hip (another instruction scheduler, see the Makefile).
#include "mfg_core.h"
namespace mfg {
int launch_core_large_mixed_ilp(const CoreArgs& a, bool sample, bool td, int num_cus, hipStream_t st);
int launch_core_large_mixed(const CoreArgs& a, bool sample, bool td, int num_cus, hipStream_t st) {
  const int R = (a.d + WAVE - 1) / WAVE;
  if (R == 2 || R == 4) return launch_core_large_mixed_ilp(a, sample, td, num_cus, st);
  return launch_core_large_impl<true, 2>(a, sample, td, num_cus, st);
}
}  // namespace mfg
