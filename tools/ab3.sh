#!/bin/bash
# usage: bash tools/ab3.sh "<lib names in csrc/variants, '' = shipped>" <probe.py> [args]   -- alternates builds, two rounds
R=$GRAFT_REPO_ROOT; LIBS=$1; shift
for rep in 1 2; do
  for v in $LIBS; do
    if [ "$v" = shipped ]; then unset MFG_HIP_LIB; else export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; fi
    echo "== $v"; python "$@" 2>&1 | grep -v "^$\|amdgpu.ids\|projected"
  done
done
