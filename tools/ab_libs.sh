#!/bin/bash
# Developer tool (GPU box): the same probes against several builds of the library (MFG_HIP_LIB), one process each, same box:
#   bash tools/ab_libs.sh <lib.so | shipped> ...
R=$GRAFT_REPO_ROOT
for L in "$@"; do
  if [ "$L" != shipped ]; then export MFG_HIP_LIB=$R/$L; else unset MFG_HIP_LIB; fi
  echo "===== library: $L"
  python3 $R/tools/step_probe.py 21,4096 21,983040 2>&1 | grep -v amdgpu.ids
  python3 $R/tools/core_probe.py 21,65536,15 21,8192,15 2>&1 | grep -v amdgpu.ids
  MFG_MAPPING=1 python3 $R/tools/shard_table.py 21 15 4096 2>&1 | grep "B="
  python3 $R/tools/irl_step_probe.py 4096 2>&1 | grep -v amdgpu.ids
  python3 $R/tools/perf_train.py 65536 2>&1 | grep -v amdgpu.ids
done
