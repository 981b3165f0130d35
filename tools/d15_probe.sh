#!/bin/bash
# Developer tool (GPU box), round 6: the packed d = 15 kernels against several builds of the library, same box:  bash tools/d15_probe.sh <lib.so | shipped> ...
R=$GRAFT_REPO_ROOT
export MFG_MAPPING=1
for L in "$@"; do
  if [ "$L" != shipped ]; then export MFG_HIP_LIB=$R/$L; else unset MFG_HIP_LIB; fi
  echo "===== library: $L (packed mapping forced)"
  python3 $R/tools/step_probe.py 15,4096 15,1966080 2>&1 | grep -v amdgpu.ids
  python3 $R/tools/shard_table.py 15 15 65536 8192 2>&1 | grep "B="
done
