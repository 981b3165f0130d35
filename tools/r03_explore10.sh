#!/bin/bash
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "large or sampler or 128 or 256 or 130 or 192 or 320 or 250 or 253 or 450 or fullsize or sweep or oracle" 2>&1 | tail -2
python $R/tools/large_probe.py 2>&1 | grep "d="
python $R/tools/large_probe.py 2>&1 | grep "TD"
