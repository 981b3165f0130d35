#!/bin/bash
# Register / scratch / spill summary per kernel of one translation unit (compile only, no GPU):
#   tools/kres.sh discrete_mean_field_game_amd/csrc/mfg_core_small.hip [extra hipcc flags]
# Uses the flags of csrc/Makefile for the sampling units (-fno-slp-vectorize, iterative-ilp for mfg_core_small).
src=$1; shift
extra=""
case "$src" in *mfg_core_small*) extra="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp";; *mfg_core_large_mixed_ilp*) extra="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp";; *mfg_core_large_mixed*) extra="-fno-slp-vectorize";; *mfg_reward_net*) extra="-fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-function -Wno-pass-failed $extra "$@" \
  -Rpass-analysis=kernel-resource-usage -c -o /dev/null "$src" 2>&1 | sed 's/ *\[-Rpass-analysis=kernel-resource-usage\]//' | awk '
  /Function Name:/ {name=$5}
  /VGPRs:/ && !/Spill/ {v=$NF}
  /ScratchSize/ {sc=$(NF)}
  /Occupancy/ {oc=$NF}
  /SGPRs Spill/ {ss=$NF}
  /VGPRs Spill/ {vs=$NF; printf "%-70s vgpr %3s occ %s scratch %4s sgpr_spill %3s vgpr_spill %3s\n", name, v, oc, sc, ss, vs}'
