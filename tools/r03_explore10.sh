#!/bin/bash
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "grad or sums" 2>&1 | tail -2
bash $R/tools/prof_any.sh $R/tools/pmc_grad.py 2>&1 | grep "k_grad_mfma_small\|k_reduce"
python $R/tools/shard_table.py 21 15 65536 8192 4096 2>&1 | grep "d="
