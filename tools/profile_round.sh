#!/bin/bash
# Run on the GPU box: bench + rocprofv3 kernel stats of the same command + PMC passes (HBM traffic of the given-P
# kernels at d = 21 / 128 / 256, SQ issue counters of the fused rollout kernels at the bench shape, C3 and the C5 share).
# Counters are collected in their own runs (never combined with the trace domains gpurun refuses).
# usage: bash tools/profile_round.sh <tag>     (outputs under gpurun_out/<tag>/; copy the summaries into profiles/)
R=$GRAFT_REPO_ROOT; TAG=${1:-r06}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
# (the CPU-baseline leg forks one process per host core; it is left out of the traced run -- the kernels are the same)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/prof.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
# round 6: the same at EXACTLY the arguments the driver runs at round end (--steps 20 --warmup 5): the line, and the
# rocprofv3 per-kernel table of that command
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_drv -o bench -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_args_under_rocprof.json 2> $O/prof_drv.err
cp $(find $O/prof_drv -name "*kernel_stats.csv" | head -1) $O/bench_driver_args_kernel_stats.csv
rm -rf $O/prof_drv
for SH in 21,15,65536 128,1,16384 256,1,16384; do
  N=${SH//,/_}
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$N -o p -- python3 $R/tools/pmc_step.py $SH > $O/pmc_fetch_$N.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$N -o p -- python3 $R/tools/pmc_step.py $SH > $O/pmc_write_$N.log 2>&1
  F=$(find $O/pmc_fetch_$N -name "*counter_collection.csv" | head -1); W=$(find $O/pmc_write_$N -name "*counter_collection.csv" | head -1)
  python3 $R/tools/summarize_pmc.py $F $W $TAG $SH > $O/pmc_traffic_$N.json
done
for SH in 21,15,65536 128,40,16384 256,40,16384; do
  N=${SH//,/_}
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq1_$N -o p -- python3 $R/tools/pmc_rollout.py $SH > $O/pmc_sq1_$N.log 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq2_$N -o p -- python3 $R/tools/pmc_rollout.py $SH > $O/pmc_sq2_$N.log 2>&1
  S1=$(find $O/pmc_sq1_$N -name "*counter_collection.csv" | head -1); S2=$(find $O/pmc_sq2_$N -name "*counter_collection.csv" | head -1)
  python3 $R/tools/summarize_sq.py $S1 $S2 $TAG $SH > $O/pmc_sq_$N.json
done
# the other configurations' kernel tables (event-free: rocprofv3 per-kernel averages)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg -o cfg -- python3 $R/tools/core_probe.py 21,4096,15 128,16384,40 256,16384,40 > $O/core_probe_under_rocprof.txt 2> $O/prof_cfg.err
cp $(find $O/prof_cfg -name "*kernel_stats.csv" | head -1) $O/other_configs_kernel_stats.csv
rm -rf $O/prof $O/prof_cfg $O/pmc_fetch_* $O/pmc_write_* $O/pmc_sq1_* $O/pmc_sq2_*
ls $O; cat $O/bench.json | head -c 600
# round 3: shard table (strong-scaling shards of the headline batch), cycle tables of the fused kernels, issue-rate micro
python3 $R/tools/shard_table.py > $O/shards.txt 2>&1
[ -x $R/tools/micro/valu_rates ] && $R/tools/micro/valu_rates > $O/valu_rates.txt 2>&1
[ -x $R/tools/micro/mfma_f64_rate ] && $R/tools/micro/mfma_f64_rate > $O/mfma_f64_rate.txt 2>&1
[ -x $R/tools/micro/mfma_valu_overlap ] && $R/tools/micro/mfma_valu_overlap > $O/mfma_valu_overlap.txt 2>&1
# (cycle tables need the ablation variant libraries of tools/ablate.sh under csrc/variants, which no longer travel to the GPU box:
#  built into MFG_VARIANT_DIR=csrc/ab when a round changes the fused kernels; round 5 did not)
VD=${MFG_VARIANT_DIR:-$R/discrete_mean_field_game_amd/csrc/ab}
[ -f $VD/libabl_PHILOX.so ] && for SH in 21,15,65536 128,40,16384 256,40,16384; do MFG_VARIANT_DIR=$VD bash $R/tools/cycle_table.sh $SH > $O/cycle_table_d${SH%%,*}.txt 2>&1; done
bash $R/tools/trace_gaps.sh $R/tools/irl_mode_probe.py 4096 > $O/irl_step_mode_trace.txt 2>&1
python3 $R/tools/perf_train.py 4096 > $O/perf_train_4096.txt 2>&1
python3 $R/tools/perf_train.py 65536 > $O/perf_train_65536.txt 2>&1
python3 $R/tools/irl_step_probe.py 4096 > $O/irl_step_probe_4096.txt 2>&1
# round 4: the matrix-core reward-network kernel: event-timed launches, phase stamps (instrumented variant library, if built:
# bash tools/variant.sh rn_stamps mfg_reward_net.hip "-DMFG_RN_STAMPS"), SQ counters at 65 536 samples
python3 $R/tools/rn_probe.py 4096 65536 > $O/rn_probe.txt 2>&1
V=$R/discrete_mean_field_game_amd/csrc/variants/librn_stamps.so
[ -f $V ] && MFG_HIP_LIB=$V python3 $R/tools/rn_stamps.py 4096 > $O/rn_stamps_4096.txt 2>&1
bash $R/tools/pmc_sq.sh k_reward_net_mfma $R/tools/rn_probe.py 65536 > $O/pmc_sq_reward_net_65536.txt 2>&1
# round 5: the IRL experiment: reward-learning kernels (event timed + rocprofv3 per-kernel table), outer-loop split, rollout-mode trace
python3 $R/tools/rn_train_probe.py > $O/rn_train_probe.txt 2>&1
bash $R/tools/prof_any.sh $R/tools/rn_train_probe.py > $O/rn_train_kernel_stats.txt 2>&1
python3 $R/tools/irl_outer_probe.py 4096 3 100 200 > $O/irl_outer_probe_4096.txt 2>&1
TG_ROWS=30 bash $R/tools/trace_gaps.sh $R/tools/irl_rollout_probe.py 4096 > $O/irl_rollout_mode_trace.txt 2>&1
cd /tmp
# round 4: the multi-rank update cycle on a 1-rank RCCL communicator (bench.py --force-dist: per-episode loop, deferred update,
# ONE all-reduce per update) at the 8-GPU shard and the full batch, with the measured latency of the exchange step
for BB in 8192 65536; do
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python3 $R/bench.py --gpus 1 --force-dist --steps 50 --warmup 10 \
    --no-configs --no-cpu-baseline --no-roofline --batch $BB > $O/collective_1rank_$BB.json 2> $O/collective_1rank_$BB.err
done
# round 6: the two lane mappings of the d = 21 sampling launches side by side (mfg_set_core_mapping: 1 packed, 2 one trajectory
# per wave): update times at the under-filled batch sizes, SQ counters of both kernels at 4 096 and 1 024 trajectories
for m in 1 2; do
  MFG_MAPPING=$m python3 $R/tools/shard_table.py 21 15 8192 6144 4096 3072 2048 1024 512 2>&1 | grep -v amdgpu.ids > $O/ab_shards_mapping$m.txt
done
( for m in 1 2; do export MFG_MAPPING=$m; for sh in 21,15,4096 21,15,1024; do echo "== lane mapping $m (1 packed k_core_small, 2 k_core_row3), shape d,T,B = $sh"; bash $R/tools/pmc_sq.sh k_core_ $R/tools/pmc_rollout.py $sh; done; done ) > $O/ab_pmc_sq.txt 2>&1
unset MFG_MAPPING
python3 $R/tools/hybrid_probe.py 3072 1024 2048 2048 > $O/hybrid_probe.txt 2>&1
