#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python tools/perf_train.py 4096
timeout 600 bash tools/trace_gaps.sh $R/tools/step_mode_probe.py 4096 | tail -8
