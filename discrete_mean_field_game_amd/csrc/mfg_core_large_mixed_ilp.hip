// Instantiations of the wave-per-trajectory core kernel, mixed precision, R = 2 and 4 (d = 65..128 and 193..256: the C3 and
// C5 shapes of BASELINE.json), built with LLVM's iterative ILP scheduler (Makefile).
#include "mfg_core.h"
namespace mfg {
int launch_core_large_mixed_ilp(const CoreArgs& a, bool sample, bool td, int num_cus, hipStream_t st) {
  return launch_core_large_impl<true, 1>(a, sample, td, num_cus, st);
}
}  // namespace mfg
