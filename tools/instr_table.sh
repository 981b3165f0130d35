#!/bin/bash
# Per-category DYNAMIC instruction table of the fused training-rollout kernels (run on the GPU box after
# `bash tools/ablate.sh build` in the build container): SQ_INSTS_VALU / SQ_WAVES of the kernel with one piece at a
# time replaced by a 1-4 instruction stand-in; baseline minus ablated = VALU instructions that piece executes per wave.
# usage: bash tools/instr_table.sh [d,T,B] > gpurun_out/<tag>/instr_table.txt
R=$GRAFT_REPO_ROOT; V=${MFG_VARIANT_DIR:-$R/discrete_mean_field_game_amd/csrc/variants}; SH=${1:-21,15,65536}
cd /tmp && export TMPDIR=/tmp
count() {  # $1 = library ('' = shipped build)
  rm -rf /tmp/it
  if [ -n "$1" ]; then export MFG_HIP_LIB=$1; else unset MFG_HIP_LIB; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/it -o p -- python3 $R/tools/pmc_rollout.py $SH > /tmp/it.log 2>&1
  python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/it/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'k_core_' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print('%.0f %.0f %.0f' % (m['SQ_INSTS_VALU'] / m['SQ_WAVES'], m['SQ_INSTS_SALU'] / m['SQ_WAVES'], m['SQ_INSTS_LDS'] / m['SQ_WAVES']))
PY
}
d=${SH%%,*}; rest=${SH#*,}; T=${rest%%,*}
read bv bs bl <<< "$(count '')"
if [ "$d" -le 64 ]; then EL=$((d * d * T / d)); else EL=$(( (d * d / 64) * T )); fi   # matrix elements per lane per launch
echo "shape d,T,B = $SH   elements per lane per launch: $EL"
echo "baseline: VALU/wave $bv ($(python3 -c "print('%.1f' % ($bv / $EL))") per element), SALU/wave $bs, LDS/wave $bl"
printf "%-10s %12s %14s %10s\n" piece "VALU/wave" "delta/wave" "per-elem"
for a in PHILOX BM SETUP TRY HTAB LNY EPI COLREW V; do
  read v s l <<< "$(count $V/libabl_$a.so)"
  printf "%-10s %12s %14s %10s\n" $a $v $((bv - v)) $(python3 -c "print('%.1f' % (($bv - $v) / $EL))")
done
