#!/bin/bash
R=$GRAFT_REPO_ROOT
for v in shipped gs_nomfma gs_noload gs_bpc4 gs_bpc1; do
  if [ "$v" = shipped ]; then unset MFG_HIP_LIB; else export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; fi
  echo "== $v"; bash $R/tools/prof_any.sh $R/tools/pmc_grad.py 2>&1 | grep "k_grad_mfma_small"
done
