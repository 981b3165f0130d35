#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
for v in "" g_u8 g_bpc3 g_bpc2 g_u2 g_u8bpc2; do
  if [ -n "$v" ]; then export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; else unset MFG_HIP_LIB; fi
  echo "== variant '$v'"
  for B in 65536 8192; do timeout 600 bash tools/prof_any.sh $R/tools/shard_table.py 21 15 $B 2>&1 | grep -E "rollout kernel|k_grad_mfma|k_reduce"; done
done > $O/grad_variants.txt 2>&1
cat $O/grad_variants.txt
