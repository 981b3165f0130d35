#!/bin/bash
R=$GRAFT_REPO_ROOT
for v in shipped t1os1 t1os4 t1os8 shipped t1os4; do
  if [ "$v" = shipped ]; then unset MFG_HIP_LIB; else export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; fi
  echo "== $v"; python $R/tools/step65k_probe.py 2>&1 | grep "step mode"; python $R/tools/step65k_probe.py 16384 2>&1 | grep "step mode"
done
