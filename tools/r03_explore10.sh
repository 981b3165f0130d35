#!/bin/bash
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "parity or classes or sampler" 2>&1 | tail -2
bash $R/tools/ab3.sh "head shipped" $R/tools/shard_table.py 21 15 65536 8192 4096 | grep "==\|d="
