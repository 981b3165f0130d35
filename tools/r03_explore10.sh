#!/bin/bash
R=$GRAFT_REPO_ROOT
for v in shipped w44 shipped w44; do
  if [ "$v" = shipped ]; then unset MFG_HIP_LIB; else export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; fi
  echo "== $v"; python $R/tools/large_probe.py 256,40,16384 2>&1 | grep "d="
done
