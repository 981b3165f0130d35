#!/bin/bash
R=$GRAFT_REPO_ROOT
bash $R/tools/ab3.sh "ts4 ts1 ts2 tsf" $R/tools/shard_table.py 21 15 65536 4096
