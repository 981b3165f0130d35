#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
tools/micro/mfma_f64_rate > $O/mfma_f64_rate.txt 2>&1
for B in 65536 8192; do timeout 600 bash tools/prof_any.sh $R/tools/shard_table.py 21 15 $B > $O/shard_kernels_$B.txt 2>&1; done
timeout 900 bash tools/cycle_table.sh 21,15,65536 > $O/cycle_table_d21.txt 2>&1
cat $O/mfma_f64_rate.txt $O/shard_kernels_65536.txt $O/shard_kernels_8192.txt $O/cycle_table_d21.txt
