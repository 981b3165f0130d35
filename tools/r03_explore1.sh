#!/bin/bash
# round-3 exploration 1 (GPU box): tests, micro rates, shard table, QUAD_WIDTH=4 A/B, phase stamps, IRL step trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 300 tools/micro/valu_rates > $O/valu_rates.txt 2>&1
timeout 600 python tools/shard_table.py > $O/shards_shipped.txt 2>&1
MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/libqw4.so timeout 600 python tools/shard_table.py > $O/shards_qw4.txt 2>&1
for B in 1024 4096 8192 65536; do MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/libtiming.so timeout 300 python tools/phase_timing.py $B; done > $O/phase_timing.txt 2>&1
timeout 600 bash tools/trace_gaps.sh $R/tools/irl_mode_probe.py 4096 > $O/irl_step_trace.txt 2>&1
timeout 600 python tools/perf_train.py 4096 > $O/perf_train_4096.txt 2>&1
cat $O/shards_shipped.txt $O/shards_qw4.txt
