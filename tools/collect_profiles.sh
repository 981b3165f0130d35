#!/bin/bash
# Copy the summaries of a tools/profile_round.sh run (gpurun_out/<tag>/) into profiles/ under the round's names.
# usage: bash tools/collect_profiles.sh <tag> [round-prefix, default r02]
O=gpurun_out/$1; P=${2:-r06}
cp $O/bench.json profiles/${P}_bench.json
cp $O/bench_kernel_stats.csv profiles/${P}_bench_kernel_stats.csv
cp $O/bench_under_rocprof.json profiles/${P}_bench_under_rocprof.json
for f in bench_driver_args.json bench_driver_args_under_rocprof.json bench_driver_args_kernel_stats.csv; do [ -f $O/$f ] && cp $O/$f profiles/${P}_$f; done
cp $O/pmc_traffic_21_15_65536.json profiles/${P}_pmc_traffic.json
cp $O/pmc_traffic_128_1_16384.json profiles/${P}_pmc_traffic_128_1_16384.json
cp $O/pmc_traffic_256_1_16384.json profiles/${P}_pmc_traffic_256_1_16384.json
cp $O/pmc_sq_21_15_65536.json profiles/${P}_pmc_sq.json
cp $O/pmc_sq_21_15_65536.json profiles/${P}_pmc_sq_21_15_65536.json
cp $O/pmc_sq_128_40_16384.json profiles/${P}_pmc_sq_128_40_16384.json
cp $O/pmc_sq_256_40_16384.json profiles/${P}_pmc_sq_256_40_16384.json
cp $O/other_configs_kernel_stats.csv profiles/${P}_other_configs_kernel_stats.csv
cp $O/core_probe_under_rocprof.txt profiles/${P}_core_probe_under_rocprof.txt
python3 - <<PY
import json
d = json.load(open('profiles/${P}_bench.json'))
r = d['roofline']
print('value %.4g env-steps/s, %.4f ms/step; given-P %.0f GB/s frac %.3f (%.1f us)' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['avg_launch_us']))
for c in d['configs']:
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in c.items() if k in ('config', 'fused_ms_per_rollout', 'fused_env_steps_per_s', 'given_P_frac', 'env_steps_per_s')})
PY
for f in ab_shards_mapping1 ab_shards_mapping2 ab_pmc_sq hybrid_probe shards valu_rates mfma_f64_rate mfma_valu_overlap cycle_table_d21 cycle_table_d128 cycle_table_d256 irl_step_mode_trace perf_train_4096 perf_train_65536 irl_step_probe_4096 rn_probe rn_stamps_4096 pmc_sq_reward_net_65536 rn_train_probe rn_train_kernel_stats irl_outer_probe_4096 irl_rollout_mode_trace; do
  [ -f $O/$f.txt ] && cp $O/$f.txt profiles/${P}_$f.txt
done
python3 - <<PY
import json
z = json.loads([l for l in open('$O/collective_1rank_8192.json') if l.startswith('{')][-1])
z2 = json.loads([l for l in open('$O/collective_1rank_65536.json') if l.startswith('{')][-1])
out = {'source': 'bench.py --gpus 1 --force-dist (1-rank RCCL communicator on one MI355X; tools/profile_round.sh)',
       'collective': z['collective'],
       'multi_rank_cycle_ms_per_update': {'batch_8192': z['ms_per_step'], 'batch_65536': z2['ms_per_step']},
       'note': 'per update: rollout kernel (previous update applied in its weight staging) | batch sums | ONE all-reduce of G, '
               'issued by the loop that loops.headline_loop names (the native RCCL loop mfg_train_rollouts_dist once its canary passed, else torch.distributed); a 1-rank collective is a no-op floor, not a latency',
       'loops_8192': z.get('loops'), 'loops_65536': z2.get('loops')}
json.dump(out, open('profiles/${P}_collective_1rank.json', 'w'), indent=1)
print(out)
PY
