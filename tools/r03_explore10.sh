#!/bin/bash
R=$GRAFT_REPO_ROOT
for v in shipped lf64w3 shipped lf64w3; do
  if [ "$v" = shipped ]; then unset MFG_HIP_LIB; else export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; fi
  echo "== $v"; python $R/tools/core_probe.py f64 192,4096,10 320,4096,5 128,8192,10 2>&1 | grep "TD rollout"
done
