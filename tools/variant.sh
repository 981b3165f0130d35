#!/bin/bash
# Developer tool: build a variant of the library with extra -D flags on ONE translation unit (no GPU needed), for A/B
# timing through MFG_HIP_LIB:   bash tools/variant.sh <name> <file.hip> "<-D flags>"   ->  csrc/variants/lib<name>.so
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/discrete_mean_field_game_amd/csrc; V=${MFG_VARIANT_DIR:-$C/variants}   # (MFG_VARIANT_DIR=$C/ab: a directory that travels with gpurun)
mkdir -p $V
NAME=$1; TU=$2; FLAGS=$3
EXTRA=""
[ "$TU" = mfg_core_small.hip ] && EXTRA="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp"
[ "$TU" = mfg_core_large_mixed.hip ] && EXTRA="-fno-slp-vectorize"
[ "$TU" = mfg_core_large_mixed_ilp.hip ] && EXTRA="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp"
[ "$TU" = mfg_reward_net.hip ] && EXTRA="-fno-slp-vectorize"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -Wno-unused-function -Wno-pass-failed $EXTRA $FLAGS -c -o $V/$NAME.o $C/$TU || exit 1
OBJS=""
for f in mfg_kernels mfg_core_small mfg_core_large_f64 mfg_core_large_mixed mfg_core_large_mixed_ilp mfg_reward_net mfg_reward_train; do
  if [ "$f.hip" = "$TU" ]; then OBJS="$OBJS $V/$NAME.o"; else OBJS="$OBJS $C/$f.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $V/lib$NAME.so $OBJS && echo $V/lib$NAME.so
