#!/bin/bash
R=$GRAFT_REPO_ROOT
for v in shipped gfold shipped gfold; do
  if [ "$v" = shipped ]; then unset MFG_HIP_LIB; else export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; fi
  echo "== $v"; python $R/tools/large_probe.py 2>&1 | grep "TD"
done
for v in shipped gfold; do
  if [ "$v" = shipped ]; then unset MFG_HIP_LIB; else export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; fi
  echo "== $v"; python $R/tools/score_error.py 128,2048,8 256,512,5 192,256,3 2>&1 | grep "d="
done
