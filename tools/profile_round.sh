#!/bin/bash
# Run on the GPU box: bench + rocprofv3 kernel stats of the same command + PMC traffic passes.
# usage: bash tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>/)
R=$GRAFT_REPO_ROOT; TAG=${1:-r01}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 10 --warmup 2 > $O/bench_under_rocprof.json 2> $O/prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python3 $R/tools/pmc_step.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python3 $R/tools/pmc_step.py > $O/pmc_write.log 2>&1
cat $O/bench.json
