#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "grad_accumulate or native or C4 or check_finite or rollout_fused or train_rollout or oracle_replay or hip_graph" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
for B in 65536 8192 4096; do timeout 600 bash tools/prof_any.sh $R/tools/shard_table.py 21 15 $B 2>&1 | grep -E "rollout kernel|k_grad_mfma|k_reduce"; done
timeout 600 python tools/perf_train.py 4096
timeout 600 bash tools/trace_gaps.sh $R/tools/irl_mode_probe.py 4096 | tail -7
