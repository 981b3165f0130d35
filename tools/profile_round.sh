#!/bin/bash
# Run on the GPU box: bench + rocprofv3 kernel stats of the same command + PMC passes (HBM traffic of the given-P
# kernel, SQ issue counters of the fused rollout kernel).  Counters are collected in their own runs.
# usage: bash tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>/; copy the summaries into profiles/)
R=$GRAFT_REPO_ROOT; TAG=${1:-r01}; O=$R/gpurun_out/$TAG; mkdir -p $O
SHAPE=21,15,65536
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --steps 10 --warmup 2 > $O/bench_under_rocprof.json 2> $O/prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python3 $R/tools/pmc_step.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python3 $R/tools/pmc_step.py > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq1 -o p -- python3 $R/tools/pmc_rollout.py > $O/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq2 -o p -- python3 $R/tools/pmc_rollout.py > $O/pmc_sq2.log 2>&1
F=$(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write -name "*counter_collection.csv" | head -1)
S1=$(find $O/pmc_sq1 -name "*counter_collection.csv" | head -1); S2=$(find $O/pmc_sq2 -name "*counter_collection.csv" | head -1)
python3 $R/tools/summarize_pmc.py $F $W $TAG $SHAPE > $O/pmc_traffic.json
python3 $R/tools/summarize_sq.py $S1 $S2 $TAG $SHAPE > $O/pmc_sq.json
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
cat $O/bench.json; cat $O/pmc_sq.json | head -60
