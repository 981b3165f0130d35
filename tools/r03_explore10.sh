#!/bin/bash
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "large or sampler or 128 or 256 or 130 or 192 or 320 or 250 or 253 or 450 or fullsize or sweep" 2>&1 | tail -2
for v in head shipped head shipped; do
  if [ "$v" = shipped ]; then unset MFG_HIP_LIB; else export MFG_HIP_LIB=$R/discrete_mean_field_game_amd/csrc/variants/lib$v.so; fi
  echo "== $v"; python $R/tools/large_probe.py 256,40,16384 192,40,4096 128,40,16384 2>&1 | grep "d="
done
