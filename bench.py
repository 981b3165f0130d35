#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec for batched d-bin population rollouts (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N == 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W    (N > 1, one rank per GPU, RCCL)

One bench "step" = one full training rollout of the forward-RL actor-critic (mfg_ac2.train semantics,
update_every='rollout') over the rank's trajectory batch, with everything resident in HBM:
    start-state gather -> fused T-step rollout kernel (Dirichlet action sampling, pi' = P^T pi,
    reward, value, TD error, score) -> batch gradient sums -> [one RCCL all-reduce of the fused
    gradient buffer when N > 1] -> (theta, w) update on device.
value = (ranks * B * T * K) / max-over-ranks wall time between barriers.  Weak scaling: the per-GPU
batch is fixed (default 65536, the north-star target point d=21, T=15).

The same JSON line carries
  roofline     : the HBM-bound given-P kernel (transition + reward over materialised actions; the fused
                 rollout never sends P through HBM), timed live with events on the
                 launch stream over B*T transitions (P slab > L3), algorithmic bytes 4(d^2+2d+1)/step;
  fused_kernel : per-launch time of the fused rollout kernel (compute bound: transcendental + RNG);
  cpu_baseline : the NumPy restatement of the reference loop (oracle/, batch 1, 1 thread) timed on
                 this box's host cores on a bounded sample (rank 0, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable copy rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--d', type=int, default=21)
    ap.add_argument('--T', type=int, default=15)
    ap.add_argument('--batch', type=int, default=65536, help='trajectories per GPU')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the CPU baseline leg')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl',
                    help="collective backend; 'gloo' (+ --share-gpu) only exists to smoke-test the N>1 path on a 1-GPU box")
    ap.add_argument('--share-gpu', action='store_true', help='debug: all ranks use GPU 0')
    ap.add_argument('--force-dist', action='store_true',
                    help='debug: take the process-group / all-reduce path even with one rank (RCCL smoke test on a 1-GPU box)')
    return ap.parse_args()


def cpu_baseline(d, budget_s):
    """Reference-faithful leg: oracle train loop, batch 1, one thread (kind 'port')."""
    from oracle import mfg_oracle as O
    # calibrate on a few episodes, then run a bounded sample
    n, t = O.cpu_baseline_steps(d, 45)
    rate = n / t
    steps = int(max(150, min(rate * budget_s, 60000)) // 15 * 15)
    n, t = O.cpu_baseline_steps(d, steps)
    out = {'value': n / t, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port',
           'sample': 'oracle.train_mfg_ac2 (NumPy restatement of mfg_ac2.train, batch 1, 1 thread): %d env-steps '
                     'of d=%d, T=15 in %.1f s on %d host cores available' % (n, d, t, os.cpu_count() or 1)}
    # all-cores mode: independent single-thread processes over disjoint trajectories
    try:
        import subprocess
        procs = max(1, min(os.cpu_count() or 1, 64))
        per = int(max(150, min(rate * 4.0, 30000)) // 15 * 15)
        code = ('import sys; sys.path.insert(0, %r); from oracle import mfg_oracle as O; '
                'n, t = O.cpu_baseline_steps(%d, %d, seed=int(sys.argv[1])); print(n, t)' % (ROOT, d, per))
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
        t0 = time.perf_counter()
        ps = [subprocess.Popen([sys.executable, '-c', code, str(100 + k)], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, env=env) for k in range(procs)]
        res = []
        for q in ps:
            try:
                o, _ = q.communicate(timeout=120)
                n_k, t_k = o.decode().split()[-2:]
                res.append((int(n_k), float(t_k)))
            except Exception:
                q.kill()
        wall = time.perf_counter() - t0
        if res:
            busy = max(r[1] for r in res)
            out['all_cores'] = {'value': sum(r[0] for r in res) / busy, 'unit': 'env-steps/s', 'cores': len(res),
                                'sample': '%d processes x %d env-steps, slowest worker %.1f s (wall %.1f s incl. start-up)'
                                          % (len(res), per, busy, wall)}
    except Exception as exc:  # the baseline is informational: never fail the bench on it
        out['all_cores'] = {'error': repr(exc)}
    return out


def pmc_traffic(kernel_prefix, d, T, B):
    """HBM bytes per launch of `kernel_prefix` from the newest committed rocprofv3 --pmc summary
    (profiles/rNN_pmc_traffic.json, written by tools/summarize_pmc.py) if it was taken at this shape."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_traffic.json')), reverse=True):
        try:
            z = json.load(open(path))
        except Exception:
            continue
        sh = z.get('shape', {})
        if (sh.get('d'), sh.get('T'), sh.get('B')) != (d, T, B):
            continue
        for name, e in z.get('kernels', {}).items():
            if name.startswith(kernel_prefix):
                return e.get('hbm_bytes_per_launch'), os.path.basename(path)
    return None, None


def pmc_sq(kernel_prefix, d, T, B):
    """VALU-busy fraction of the fused kernel from the newest committed SQ counter summary
    (profiles/rNN_pmc_sq.json, written by tools/summarize_sq.py) taken at this shape."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_sq.json')), reverse=True):
        try:
            z = json.load(open(path))
        except Exception:
            continue
        sh = z.get('shape', {})
        if (sh.get('d'), sh.get('T'), sh.get('B')) != (d, T, B):
            continue
        best = None
        for name, e in z.get('kernels', {}).items():
            if name.replace('mfg::', '').startswith(kernel_prefix) and 'valu_busy' in e:
                if best is None or e.get('dur_us_under_pmc', 0) > best[1].get('dur_us_under_pmc', 0):
                    best = (name, e)
        if best:
            return {'valu_busy': best[1]['valu_busy'], 'valu_insts_per_wave': best[1].get('valu_insts_per_wave'),
                    'pmc_source': os.path.basename(path)}
    return {}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist
    from discrete_mean_field_game_amd import ops, _lib

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        sys.exit('launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU: the HIP path has no CPU fallback')
    _lib.lib()  # fail loudly if the extension is missing
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29517')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)

    d, T = args.d, args.T
    B = args.batch if args.scaling == 'weak' else args.batch // world
    F = ops.num_features(d)
    theta0, shift, alpha_scale, gamma = 8.86349, 0.16, 12000.0, 1.0      # mfg_ac2.py:832
    lr_c, lr_a = 0.1, 0.001                                              # mfg_ac2.py:448

    # synthetic workload: 64 Dirichlet(1) start states rounded through '%.3e' text
    rs = np.random.RandomState(0)
    mat = rs.dirichlet(np.ones(d), size=64)
    mat = np.array([[float('%.3e' % v) for v in row] for row in mat], dtype=np.float32)
    mat_pi0 = torch.as_tensor(mat, device=dev)
    idx = torch.as_tensor(np.random.RandomState(1234 + rank).randint(64, size=B).astype(np.int32), device=dev)
    w = torch.as_tensor(np.random.RandomState(1).rand(F), device=dev)    # U[0,1)^F, mfg_ac2.py:176
    theta = torch.tensor([theta0], dtype=torch.float64, device=dev)
    G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
    ws = ops.workspace(B * T, d, dev)
    bufs = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'reward': torch.empty(B, T, device=dev),
            'delta': torch.empty(B, T, dtype=torch.float64, device=dev),
            'g': torch.empty(B, T, dtype=torch.float64, device=dev)}
    traj_offset = rank * B

    def all_reduce_(t, op=dist.ReduceOp.SUM):
        if args.backend == 'gloo':                                       # debug path: stage through the host
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def one_step(k):
        sc = 1.0 / (k + 1)
        sa = 1.0 / ((k + 1) * np.log(np.log(k + 20)))                    # mfg_ac2.py:514,522
        # start-state gather (in-kernel) + fused rollout + batch sums; on one GPU the update rides in the same call
        ops.train_rollout(mat_pi0, idx, T, theta, shift, alpha_scale, w, gamma, G, ws, bufs, lr_c * sc, lr_a * sa,
                          apply=not multi, seed=2024, first_step=k * T, traj_offset=traj_offset)
        if multi:
            all_reduce_(G)                                               # one RCCL all-reduce per update
            ops.apply_update(G, d, lr_c * sc, lr_a * sa, w, theta)

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        one_step(k)
    sync()
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + args.steps):
        one_step(k)
    sync()
    elapsed = time.perf_counter() - t0
    if multi:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce_(te, op=dist.ReduceOp.MAX)
        elapsed = float(te[0])
    theta_end = float(theta[0])
    if not np.isfinite(theta_end):
        sys.exit('non-finite theta after the timed region')
    value = world * B * T * args.steps / elapsed

    def event_time(fn, n, warm=2):
        for _ in range(warm):
            fn()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / n

    out = None
    if rank == 0:
        bytes_per_step = 4 * (d * d + 2 * d + 1)                         # SURVEY.md 8d / BASELINE.md 3
        pi0 = ops.gather_start(mat_pi0, idx)
        roofline = None
        fused = None
        if not args.no_roofline:
            # materialise the actions of one rollout (B*T transitions, > L3 at the default size)
            N = B * T
            r = ops.rollout(pi0, T, theta, shift, alpha_scale, seed=7, traj_offset=traj_offset, td=False, write_P=True)
            P_all = r['P'].view(N, d, d)
            pi_all = r['pi_traj'][:, :T].contiguous().view(N, d)
            # 20 untimed launches first: the kernel's first ~20 launches on a fresh slab run up to 1.5x slower (clock /
            # TLB ramp: 484 -> 337 -> 327 us per launch over the first three batches of 20)
            n_launch = 40
            t_step = event_time(lambda: ops.step_given_P(pi_all, P_all), n=n_launch, warm=20)
            achieved = N * bytes_per_step / t_step / 1e9
            traffic, traffic_src = pmc_traffic('k_step_', d, T, B)
            roofline = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                        'kernel': 'k_step_small' if d <= 64 else ('k_step_rows' if d in (128, 256) else 'k_step_large'),
                        'leg': 'given-P transition+reward over %d transitions (P slab %.2f GB)' % (N, N * d * d * 4 / 1e9),
                        'algorithmic_bytes_per_launch': N * bytes_per_step, 'avg_launch_us': t_step * 1e6,
                        'env_steps_per_s': N / t_step}
            del P_all, pi_all, r
            # this box's own streaming ceilings (stock torch kernels on a 1.6 GB buffer), for context
            xx = torch.empty(400_000_000, device=dev).uniform_()
            yy = torch.empty_like(xx)
            t_rd = event_time(lambda: xx.sum(), n=5)
            t_cp = event_time(lambda: yy.copy_(xx), n=5)
            roofline['box_ceiling_GBs'] = {'torch_sum_read': xx.numel() * 4 / t_rd / 1e9,
                                           'torch_copy_read_plus_write': 2 * xx.numel() * 4 / t_cp / 1e9}
            del xx, yy
            t_f = event_time(lambda: ops.rollout(pi0, T, theta, shift, alpha_scale, w=w, gamma=gamma, seed=7,
                                                 traj_offset=traj_offset, td=True, G=G, ws=ws, out=bufs), n=10, warm=3)
            fused = {'kernel': ('k_core_small' if d <= 64 else 'k_core_large') + '<SAMPLE,TD,MIXED> + k_grad + k_reduce_partials',
                     'bound': 'valu/transcendental+rng (P stays on chip)', 'avg_launch_ms': t_f * 1e3,
                     'env_steps_per_s': B * T / t_f, 'hbm_algorithmic_GBs': B * T * bytes_per_step / t_f / 1e9}
            fused.update(pmc_sq('k_core_', d, T, B))
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(d, args.cpu_seconds)
        out = {
            'metric': 'env-steps/sec for batched d-bin population rollouts', 'value': value, 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32 (f64 accumulate)',
            'data': 'synthetic',
            'config': {'workload': 'forward-RL actor-critic training rollouts (mfg_ac2.train, update per rollout): '
                                   'd=%d topics, T=%d, batch=%d trajectories per GPU' % (d, T, B),
                       'd': d, 'T': T, 'batch_per_gpu': B, 'global_batch': B * world,
                       'theta0': theta0, 'shift': shift, 'alpha_scale': alpha_scale, 'rng': 'philox4x32-10',
                       'parallelism': 'trajectory-sharded x%d, 1 all-reduce/update' % world},
            'theta_end': theta_end,
            'roofline': roofline, 'fused_kernel': fused, 'cpu_baseline': cpu,
        }
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
