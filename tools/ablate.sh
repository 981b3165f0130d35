#!/bin/bash
# Timing ablations of the small-d fused kernel (developer tool; results are WRONG by construction, only the time counts).
# Build (no GPU needed):   bash tools/ablate.sh build
# Run on the GPU box:      bash tools/ablate.sh run [d,B,T]
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/discrete_mean_field_game_amd/csrc; V=${MFG_VARIANT_DIR:-$C/variants}
ABL="PHILOX BM SETUP TRY HTAB LNY EPI COLREW V COLT TSUM"
if [ "$1" = build ]; then
  mkdir -p $V
  for a in $ABL; do
    ( ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-pass-failed -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -DMFG_ABL_$a -c -o $V/abl_$a.o $C/mfg_core_small.hip 2>/dev/null ||
        /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-pass-failed -fno-slp-vectorize -DMFG_ABL_$a -c -o $V/abl_$a.o $C/mfg_core_small.hip ) &&
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-pass-failed -fno-slp-vectorize -DMFG_ABL_$a -c -o $V/abll_$a.o $C/mfg_core_large_mixed.hip &&
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-pass-failed -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp -DMFG_ABL_$a -c -o $V/ablli_$a.o $C/mfg_core_large_mixed_ilp.hip &&
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $V/libabl_$a.so $C/mfg_kernels.o $V/abl_$a.o $C/mfg_core_row3.o $C/mfg_core_large_f64.o $V/abll_$a.o $V/ablli_$a.o $C/mfg_reward_net.o $C/mfg_reward_train.o ) &
    if (( $(jobs -r | wc -l) >= 5 )); then wait -n; fi
  done
  wait
  ls $V
else
  S=${2:-21,65536,15}
  echo "baseline"; python3 $R/tools/core_probe.py $S 2>/dev/null | grep -E "rollout"
  for a in $ABL; do echo "without $a"; MFG_HIP_LIB=$V/libabl_$a.so python3 $R/tools/core_probe.py $S 2>/dev/null | grep -E "rollout"; done
fi
