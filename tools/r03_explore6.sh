#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
for v in "" g_bpc1 g_bpc3 g_bpc4; do
  if [ -n "$v" ]; then export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; else unset MFG_HIP_LIB; fi
  echo "== variant '$v'"
  for B in 65536 8192; do timeout 600 bash tools/prof_any.sh $R/tools/shard_table.py 21 15 $B 2>&1 | grep -E "rollout kernel|k_grad_mfma|k_reduce"; done
done
unset MFG_HIP_LIB
timeout 600 python tools/perf_train.py 4096
timeout 600 bash tools/trace_gaps.sh $R/tools/irl_mode_probe.py 4096 | tail -24
