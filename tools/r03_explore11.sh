#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O
bash $R/tools/cycle_table.sh 128,40,16384 > $O/cycle_table_d128.txt 2>&1
head -24 $O/cycle_table_d128.txt
