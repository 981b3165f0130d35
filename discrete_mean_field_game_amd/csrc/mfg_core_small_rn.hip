// Instantiations of the packed step kernel WITH the IRL reward network inside (k_core_small<..., RN>, mfg_rn_fused.h), d = 21
// and 15, mixed precision, with and without the per-tile batch sums.  Own translation unit: built with the default scheduler
// (csrc/Makefile), see the note in mfg_core_small.hip.
#include <atomic>

#include "mfg_core.h"

namespace mfg {

template <int D, bool SUMS>
static void go_rn(const CoreArgs& a, int num_cus, size_t lds, hipStream_t st) {
  static std::atomic<size_t> cached_lds[64];
  static std::atomic<int> cached_bpc[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cached_lds[dev].load() != lds + 1) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_core_small<true, true, true, D, SUMS, true>, BLOCK, lds) != hipSuccess || n < 1)
      n = 1;
    cached_bpc[dev].store(n);
    cached_lds[dev].store(lds + 1);
  }
  const int G = WAVE / a.d, TB = WAVES * G;
  const int grid = core_grid(a.B, TB, cached_bpc[dev].load() * (a.T == 1 ? 2 : MFG_CORE_OVERSUBSCRIBE), num_cus);
  hipLaunchKernelGGL((k_core_small<true, true, true, D, SUMS, true>), dim3(grid), dim3(BLOCK), lds, st, a);
}

void launch_core_small_rn(const CoreArgs& a, int num_cus, size_t lds, hipStream_t st) {
  if (a.d == 21) {
    if (a.part_rows) go_rn<21, true>(a, num_cus, lds, st);
    else go_rn<21, false>(a, num_cus, lds, st);
  } else {
    if (a.part_rows) go_rn<15, true>(a, num_cus, lds, st);
    else go_rn<15, false>(a, num_cus, lds, st);
  }
}

}  // namespace mfg
