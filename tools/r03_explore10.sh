#!/bin/bash
R=$GRAFT_REPO_ROOT
bash $R/tools/ab3.sh "shipped expboth" $R/tools/shard_table.py 21 15 65536 8192 4096
for v in shipped expboth; do
  if [ "$v" = shipped ]; then unset MFG_HIP_LIB; else export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; fi
  echo "== $v"; python $R/tools/large_probe.py 2>&1 | grep "d="
done
