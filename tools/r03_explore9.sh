#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O; rm -f $O/grad.txt
for v in base bpc3 bpc4; do
  if [ $v = base ]; then unset MFG_HIP_LIB; else export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; fi
  echo "== $v" >> $O/grad.txt
  bash $R/tools/prof_any.sh $R/tools/large_probe.py 128,40,16384 2>&1 | grep "k_grad_mfma\|training" >> $O/grad.txt
done
cat $O/grad.txt
