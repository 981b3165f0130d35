#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "grad_accumulate or native or C4 or separable or check_finite or rollout_fused or train_rollout or retraces or oracle_replay or multiproc or hip_graph" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 600 python tools/shard_table.py > $O/shards.txt 2>&1
timeout 600 python tools/perf_train.py 4096 > $O/perf_train_4096.txt 2>&1
timeout 600 bash tools/trace_gaps.sh $R/tools/irl_mode_probe.py 4096 > $O/irl_step_trace.txt 2>&1
timeout 600 bash tools/trace_gaps.sh $R/tools/step_mode_probe.py 4096 > $O/ac_step_trace.txt 2>&1
timeout 600 bash tools/prof_any.sh $R/tools/shard_table.py 21 15 65536 8192 > $O/shard_kernels.txt 2>&1
cat $O/shards.txt $O/perf_train_4096.txt; tail -12 $O/irl_step_trace.txt; tail -8 $O/ac_step_trace.txt; cat $O/shard_kernels.txt
