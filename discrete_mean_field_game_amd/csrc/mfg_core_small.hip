// Instantiations of the packed small-d core kernel (d <= 64): generic runtime d plus compile-time
// specialisations for the reference's two problem sizes (d = 21: mfg_ac2.py:25, d = 15: ac_irl.py:33).
#include "mfg_core.h"

namespace mfg {

template <bool SAMPLE, bool TD, bool FAST, int D>
static void go(const CoreArgs& a, int grid, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((k_core_small<SAMPLE, TD, FAST, D>), dim3(grid), dim3(BLOCK), lds, st, a);
}

template <int D>
static void dispatch(const CoreArgs& a, bool sample, bool td, bool fast, int grid, size_t lds, hipStream_t st) {
  if (fast) {
    if (sample && td) go<true, true, true, D>(a, grid, lds, st);
    else if (sample) go<true, false, true, D>(a, grid, lds, st);
    else go<false, true, true, D>(a, grid, lds, st);
  } else {
    if (sample && td) go<true, true, false, D>(a, grid, lds, st);
    else if (sample) go<true, false, false, D>(a, grid, lds, st);
    else go<false, true, false, D>(a, grid, lds, st);
  }
}

static int blocks_for(size_t lds, int d, int num_cus, int64_t B) {
  const int G = WAVE / d, TB = WAVES * G;
  int bpc = (int)((160 * 1024) / (lds + 256));
  if (bpc > 8) bpc = 8;
  if (bpc < 1) bpc = 1;
  return core_grid(B, TB, bpc, num_cus);
}

int launch_core_small(const CoreArgs& a, bool sample, bool td, bool fast, int num_cus, hipStream_t st) {
  const int d = a.d;
  const bool want_v = td && a.w != nullptr;
  const size_t lds = core_small_lds(d, want_v);
  const int grid = blocks_for(lds, d, num_cus, a.B);
  if (d == 21) dispatch<21>(a, sample, td, fast, grid, lds, st);
  else if (d == 15) dispatch<15>(a, sample, td, fast, grid, lds, st);
  else dispatch<0>(a, sample, td, fast, grid, lds, st);
  return MFG_OK;
}

}  // namespace mfg
