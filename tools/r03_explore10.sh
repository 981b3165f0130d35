#!/bin/bash
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "reward_net or irl or IRL or dropout or C4" 2>&1 | tail -2
bash $R/tools/ab3.sh "head shipped" $R/tools/rn_probe.py 61440 4096 4097 8192 | grep "==\|reward"
for v in head shipped; do
  if [ "$v" = shipped ]; then unset MFG_HIP_LIB; else export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; fi
  echo "== $v"; python $R/tools/perf_train.py 4096 2>&1 | grep "ac_irl"
done
