#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O; rm -f $O/grad.txt
python -m pytest tests -m gpu -x -q -k "grad or td_pg or fullsize or large or 128 or 256" 2>&1 | tail -5
bash $R/tools/prof_any.sh $R/tools/large_probe.py 2>&1 | grep "k_grad_mfma\|k_value_mfma\|k_reduce\|training" >> $O/grad.txt
cat $O/grad.txt
