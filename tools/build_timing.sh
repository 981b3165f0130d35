#!/bin/bash
# Developer tool: the -DMFG_TIMING build of the library (s_memtime stamps in the small-d fused kernel: tools/phase_timing.py,
# tools/phase_timing_t1.py).  CoreArgs gains a debug pointer, so EVERY translation unit is rebuilt with the flag.
#   bash tools/build_timing.sh   ->  csrc/variants/libtiming.so   (MFG_VARIANT_DIR=.../csrc/ab: a directory that travels with gpurun)
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/discrete_mean_field_game_amd/csrc; V=${MFG_VARIANT_DIR:-$C/variants}; mkdir -p $V/timing
for f in mfg_kernels mfg_core_small mfg_core_large_f64 mfg_core_large_mixed mfg_core_large_mixed_ilp mfg_reward_net mfg_reward_train; do
  EX=""
  case $f in
    mfg_core_small|mfg_core_large_mixed_ilp) EX="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp";;
    mfg_core_large_mixed|mfg_reward_net) EX="-fno-slp-vectorize";;
  esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-function -Wno-pass-failed $EX -DMFG_TIMING \
    -c -o $V/timing/$f.o $C/$f.hip &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $V/libtiming.so $V/timing/*.o && echo $V/libtiming.so
