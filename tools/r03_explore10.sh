#!/bin/bash
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "parity or classes" 2>&1 | tail -2
for v in head shipped head shipped; do
  if [ "$v" = shipped ]; then unset MFG_HIP_LIB; else export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; fi
  echo "== $v"; python $R/tools/step65k_probe.py 2>&1 | grep "step mode"; python $R/tools/step65k_probe.py 4096 2>&1 | grep "step mode"
done
bash $R/tools/ab3.sh "head shipped" $R/tools/shard_table.py 21 15 65536 4096 | grep "==\|d="
