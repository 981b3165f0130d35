// Instantiations of the packed small-d core kernel (d <= 64), both precisions.
#include "mfg_core.h"

namespace mfg {

template <bool SAMPLE, bool TD, bool FAST>
static void go(const CoreArgs& a, int grid, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((k_core_small<SAMPLE, TD, FAST>), dim3(grid), dim3(BLOCK), lds, st, a);
}

int launch_core_small(const CoreArgs& a, bool sample, bool td, bool fast, int num_cus, hipStream_t st) {
  const int d = a.d;
  const bool want_v = td && a.w != nullptr;
  const size_t lds = core_small_lds(d, want_v);
  const int G = WAVE / d, TB = WAVES * G;
  int bpc = (int)((160 * 1024) / (lds + 256));
  if (bpc > 8) bpc = 8;
  if (bpc < 1) bpc = 1;
  const int grid = core_grid(a.B, TB, bpc, num_cus);
  if (fast) {
    if (sample && td) go<true, true, true>(a, grid, lds, st);
    else if (sample) go<true, false, true>(a, grid, lds, st);
    else go<false, true, true>(a, grid, lds, st);
  } else {
    if (sample && td) go<true, true, false>(a, grid, lds, st);
    else if (sample) go<true, false, false>(a, grid, lds, st);
    else go<false, true, false>(a, grid, lds, st);
  }
  return MFG_OK;
}

}  // namespace mfg
