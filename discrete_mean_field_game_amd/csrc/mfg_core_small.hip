// Instantiations of the packed small-d core kernel (d <= 64): generic runtime d plus compile-time
// specialisations for the reference's two problem sizes (d = 21: mfg_ac2.py:25, d = 15: ac_irl.py:33).
#include <atomic>

#include "mfg_core.h"

namespace mfg {

// Grid = tiles, capped at MFG_CORE_OVERSUBSCRIBE x the blocks that can be resident (registers AND LDS, asked from the
// runtime per instantiation); blocks loop over tiles beyond that.  Measured at d=21, B=65536 (5 462 tiles, 1 024
// resident blocks): exactly-resident persistent grid 2.30 ms, x1.5 2.21, x2 2.14, x4 2.07, one tile per block 2.07 --
// tiles do not take equal time (rejection retries), so the hardware dispatcher back-filling finished blocks beats
// a static tile split.
template <bool SAMPLE, bool TD, bool FAST, int D, bool SUMS = false, int STEP = 0>
static void go(const CoreArgs& a, int num_cus, size_t lds, hipStream_t st) {
  // occupancy of this instantiation at this LDS size, cached per device
  static std::atomic<size_t> cached_lds[64];
  static std::atomic<int> cached_bpc[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cached_lds[dev].load() != lds + 1) {  // (+1: zero-initialised slots never match)
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_core_small<SAMPLE, TD, FAST, D, SUMS, STEP>, BLOCK, lds) != hipSuccess || n < 1)
      n = 1;
    cached_bpc[dev].store(n);
    cached_lds[dev].store(lds + 1);
  }
  const int G = WAVE / a.d, TB = WAVES * G;
  // (single-step launches: x2 -- a block's weight staging and first state load are then shared by ~3-4 tiles; measured
  //  1.64 -> 1.61 ms per 15-step episode of per-step updates at B = 65 536, x1 1.70, x4 1.62)
  const int grid = core_grid(a.B, TB, cached_bpc[dev].load() * (a.T == 1 ? 2 : MFG_CORE_OVERSUBSCRIBE), num_cus);
  // (STEP: the blocks that reduce the previous env step's partial rows ride behind the sampling blocks)
  hipLaunchKernelGGL((k_core_small<SAMPLE, TD, FAST, D, SUMS, STEP>), dim3(grid + (STEP == 1 ? core_step_red_blocks(a.d * (a.d + 1) / 2 + a.d + 1 + 3) : 0)), dim3(BLOCK), lds, st, a);
}

template <int D>
static void dispatch(const CoreArgs& a, bool sample, bool td, bool fast, int num_cus, size_t lds, hipStream_t st) {
  if constexpr (D > 0) {
    // IRL env step (mfg_train_episode_irl): theta from the previous step's partial rows, their reduction in the same launch
    if (sample && td && a.step_nrows > 0) {
      if (fast) go<true, true, true, D, false, 1>(a, num_cus, lds, st);
      else go<true, true, false, D, false, 1>(a, num_cus, lds, st);
      return;
    }
    if (sample && td && a.step_nrows < 0) {
      if (fast) go<true, true, true, D, false, 2>(a, num_cus, lds, st);
      else go<true, true, false, D, false, 2>(a, num_cus, lds, st);
      return;
    }
    // per-step updates: the variant that also leaves the tile's batch sums (launch_core_sums in mfg_kernels.hip)
    if (sample && td && a.part_rows) {
      if (fast) go<true, true, true, D, true>(a, num_cus, lds, st);
      else go<true, true, false, D, true>(a, num_cus, lds, st);
      return;
    }
  }
  if (fast) {
    if (sample && td) go<true, true, true, D>(a, num_cus, lds, st);
    else if (sample) go<true, false, true, D>(a, num_cus, lds, st);
    else go<false, true, true, D>(a, num_cus, lds, st);
  } else {
    if (sample && td) go<true, true, false, D>(a, num_cus, lds, st);
    else if (sample) go<true, false, false, D>(a, num_cus, lds, st);
    else go<false, true, false, D>(a, num_cus, lds, st);
  }
}

int launch_core_small(const CoreArgs& a, bool sample, bool td, bool fast, int num_cus, hipStream_t st) {
  const int d = a.d;
  // d = 21 batches that under-fill the machine: one trajectory per wave, three lanes per matrix row (mfg_core_row3.hip) --
  // the same bits as the packed kernel below, a wave's serial chain ~2.2x shorter
  if (core_row3_wanted(a, sample, td, fast, num_cus)) return launch_core_row3(a, td, num_cus, st);
  const bool want_v = td && a.w != nullptr;
  const size_t lds = core_small_lds(d, want_v, sample);
  if (d == 21) dispatch<21>(a, sample, td, fast, num_cus, lds, st);
  else if (d == 15) dispatch<15>(a, sample, td, fast, num_cus, lds, st);
  else dispatch<0>(a, sample, td, fast, num_cus, lds, st);
  return MFG_OK;
}

}  // namespace mfg
