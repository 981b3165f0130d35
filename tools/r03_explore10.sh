#!/bin/bash
R=$GRAFT_REPO_ROOT
bash $R/tools/ab3.sh "head shipped oldfc4 nofc3 nofc3oldfc4" $R/tools/rn_probe.py 61440 | head -12
