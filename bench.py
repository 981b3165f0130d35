#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec for batched d-bin population rollouts (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N == 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W    (N > 1, one rank per GPU, RCCL)

One bench "step" = one episode of the drop-in class, `actor_critic.train(update_every='rollout')` -- the timed region IS a
call of the reference's train() surface (mfg_ac2.py:448) -- over the rank's trajectory batch, everything resident in HBM:
    per-episode start-state draw (on the device, fresh every episode) -> fused T-step rollout kernel (Dirichlet
    action sampling, pi' = P^T pi, reward, value, TD error, score) -> batch gradient sums -> [one RCCL all-reduce
    of the fused gradient buffer when N > 1] -> (theta, w) update on device.
value = (ranks * B * T * K) / max-over-ranks wall time between barriers.  STRONG scaling by default: the global
batch (65536, the north-star target point d=21, T=15) is split over the N GPUs; for N > 1 the same line also carries
the weak-scaling leg (65536 per GPU, `other_scaling`) and BASELINE config 5 (d=256, T=40, 131072 split N ways,
`c5_strong`); for N == 1 a bounded `configs` array covers C2 / C3 / C4 / the C5 share / the f64 headline.

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
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10,
                    help='untimed steps first (the device needs ~10 ms of this kernel to reach its sustained clock: 3 warm-up steps '
                         'measured 1.13 ms per step, 10 and more 1.08)')
    ap.add_argument('--d', type=int, default=21)
    ap.add_argument('--T', type=int, default=15)
    ap.add_argument('--batch', type=int, default=65536,
                    help='GLOBAL batch, split over the GPUs (strong scaling, default); per-GPU batch with --scaling weak')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='strong')
    ap.add_argument('--no-configs', action='store_true', help='skip the other BASELINE.json configurations (N == 1)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the CPU baseline leg')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl',
                    help="collective backend; 'gloo' (+ --share-gpu) only exists to smoke-test the N>1 path on a 1-GPU box")
    ap.add_argument('--share-gpu', action='store_true', help='debug: all ranks use GPU 0')
    ap.add_argument('--c5-batch', type=int, default=131072, help='GLOBAL batch of the c5_strong leg (N > 1); tests shrink it')
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
        procs = max(1, os.cpu_count() or 1)              # one single-thread process per host core of this box
        # time bounded: every worker repeats 150-step samples until ~6 s have passed (the cores of a shared box are not
        # all ours; a fixed step count per worker took 160 s with 255 workers where 64 workers needed 8 s)
        code = ('import sys, time; sys.path.insert(0, %r); from oracle import mfg_oracle as O\n'
                'n = 0; t = 0.0; k = 0\n'
                'while t < 6.0:\n'
                '    a, b = O.cpu_baseline_steps(%d, 150, seed=int(sys.argv[1]) + 1000 * k); n += a; t += b; k += 1\n'
                'print(n, t)' % (ROOT, d))
        env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1', MKL_NUM_THREADS='1')
        t0 = time.perf_counter()
        ps = [subprocess.Popen([sys.executable, '-c', code, str(100 + k)], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, env=env) for k in range(procs)]
        res = []
        for q in ps:
            try:
                o, _ = q.communicate(timeout=90)
                n_k, t_k = o.decode().split()[-2:]
                res.append((int(n_k), float(t_k)))
            except Exception:
                q.kill()
        wall = time.perf_counter() - t0
        if res:
            busy = max(r[1] for r in res)
            out['all_cores'] = {'value': sum(r[0] / r[1] for r in res), 'unit': 'env-steps/s', 'cores': len(res),
                                'sample': '%d single-thread processes (os.cpu_count() = %d), each repeating 150-step samples for ~6 s: '
                                          '%d env-steps in total, slowest worker %.1f s (wall %.1f s incl. start-up)'
                                          % (len(res), procs, sum(r[0] for r in res), busy, wall)}
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

    theta0, shift, alpha_scale, gamma = 8.86349, 0.16, 12000.0, 1.0      # mfg_ac2.py:832
    lr_c, lr_a = 0.1, 0.001                                              # mfg_ac2.py:448

    def all_reduce_(t, op=dist.ReduceOp.SUM):
        if args.backend == 'gloo':                                       # debug path: stage through the host
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op)

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    def start_table(d):
        # synthetic workload: 64 Dirichlet(1) start states rounded through '%.3e' text
        return torch.as_tensor(start_table64(d).astype(np.float32), device=dev)

    def start_table64(d):
        rs = np.random.RandomState(0)
        mat = rs.dirichlet(np.ones(d), size=64)
        return np.array([[float('%.3e' % v) for v in row] for row in mat])

    def make_class(d, T, B_global, mode='rollout', precision='mixed'):
        """The drop-in class on this rank: GLOBAL batch (the class shards it over the ranks of the default process group),
        reference hyper-parameters (mfg_ac2.py:832), synthetic start table, w ~ U[0,1)^F (mfg_ac2.py:176)."""
        from discrete_mean_field_game_amd.mfg_ac2 import actor_critic
        np.random.seed(1)                                                # init_w draws from np.random
        ac = actor_critic(theta=theta0, shift=shift, alpha_scale=alpha_scale, d=d, pi0=start_table64(d), batch=B_global,
                          rng='philox', seed=2024, update_every=mode, precision=precision, episode_steps=T, verbose=0,
                          device=dev)
        ac._force_collective = bool(args.force_dist)
        return ac

    def training_leg(d, T, B, steps, warmup, precision='mixed', mode='rollout', native_rccl=True):
        """`steps` timed episodes of actor_critic.train() (one update per episode in mode 'rollout', the reference's
        per-step updates in mode 'step') with B trajectories PER RANK.  native_rccl=False (several ranks): keep the exchange in
        torch.distributed (one all-reduce per update issued from the class's Python loop) even where the library's own RCCL
        loop passed its canary.  Returns (max-over-ranks seconds, theta at the end, the class instance)."""
        ac = make_class(d, T, B * world, mode, precision)
        ac._use_native_rccl = bool(native_rccl)
        ac.train(num_episodes=warmup, gamma=gamma, constant=0, lr_critic=lr_c, lr_actor=lr_a, consecutive=10 ** 9)
        sync()
        t0 = time.perf_counter()
        ac.train(num_episodes=steps, gamma=gamma, constant=0, lr_critic=lr_c, lr_actor=lr_a, consecutive=10 ** 9,
                 first_episode=warmup)
        sync()
        elapsed = time.perf_counter() - t0
        if multi:
            te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            all_reduce_(te, op=dist.ReduceOp.MAX)
            elapsed = float(te[0])
        theta_end = float(np.ravel(ac.theta)[0])
        if not np.isfinite(theta_end):
            sys.exit('non-finite theta after the timed region')
        return elapsed, theta_end, ac

    def native_leg(d, T, B, steps, warmup, precision='mixed'):
        """The same `steps` updates issued by ONE native call (mfg_train_rollouts: no interpreter between episodes), timed with
        events on the launch stream: the GPU time of the kernels alone.  class wall time / this = how much of the class
        API's time the GPU is busy (1.0 = not host bound).  Single rank only."""
        F = ops.num_features(d)
        mat = torch.as_tensor(start_table64(d).astype(np.float32), device=dev)
        w = torch.as_tensor(np.random.RandomState(1).rand(F), device=dev)
        theta = torch.tensor([theta0], dtype=torch.float64, device=dev)
        G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
        ws = ops.workspace(B * T, d, dev)
        bufs = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'pi_last': torch.empty(B, d, device=dev),
                'reward': torch.empty(B, T, device=dev), 'delta': torch.empty(B, T, dtype=torch.float64, device=dev),
                'g': torch.empty(B, T, dtype=torch.float64, device=dev)}
        racc = torch.zeros(warmup + steps, dtype=torch.float64, device=dev)
        run = lambda k, e0: ops.train_rollouts(mat, T, k, e0, False, theta, shift, alpha_scale, w, gamma, G, ws, bufs, lr_c, lr_a,
                                               seed=2024, first_step=e0 * T, reward_acc=racc.data_ptr() + 8 * e0,
                                               precision=precision)
        run(warmup, 0)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        run(steps, warmup)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3

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

    def given_p_leg(d, B, T_mat, n_launch=40, warm=20):
        """HBM-bound given-P kernel (transition + reward) over the materialised actions of T_mat steps of B
        trajectories; live event timing on the launch stream."""
        mat_pi0 = start_table(d)
        idx = torch.as_tensor(np.random.RandomState(99).randint(64, size=B).astype(np.int32), device=dev)
        pi0 = ops.gather_start(mat_pi0, idx)
        theta = torch.tensor([theta0], dtype=torch.float64, device=dev)
        r = ops.rollout(pi0, T_mat, theta, shift, alpha_scale, seed=7, td=False, write_P=True)
        N = B * T_mat
        P_all = r['P'].view(N, d, d)
        pi_all = r['pi_traj'][:, :T_mat].contiguous().view(N, d)
        t_step = event_time(lambda: ops.step_given_P(pi_all, P_all), n=n_launch, warm=warm)
        bps = 4 * (d * d + 2 * d + 1)
        return {'transitions': N, 'slab_GB': N * d * d * 4 / 1e9, 'avg_launch_us': t_step * 1e6,
                'achieved_GBs': N * bps / t_step / 1e9, 'frac': N * bps / t_step / 1e9 / HBM_PEAK_GBS,
                'env_steps_per_s': N / t_step}

    d, T = args.d, args.T
    # clock ramp: untimed launches of the WORKLOAD'S OWN kernels on every CU before anything is measured -- a separate instance of
    # the drop-in class runs MFG_BENCH_HEAT (default 40) episodes of the headline shape (~40 ms at the default size; a cold
    # device takes the first launches to leave its idle power state, and a streaming multiply in front of a VALU-bound kernel
    # settles at another clock than the kernel itself).  Not a step of the workload: the W warm-up steps below still run as asked.
    heat_eps = int(os.environ.get('MFG_BENCH_HEAT', '40'))
    if heat_eps > 0:
        Bh = args.batch // world if args.scaling == 'strong' else args.batch
        _h = make_class(d, T, Bh * world)
        _h._use_native_rccl = False          # (the heat-up must not depend on the native loop's canary having run)
        _h.train(num_episodes=heat_eps, gamma=gamma, constant=0, lr_critic=lr_c, lr_actor=lr_a, consecutive=10 ** 9)
        torch.cuda.synchronize()
        del _h

    # ---- headline leg.  strong scaling (default): the GLOBAL batch --batch is split over the ranks; weak: --batch per GPU
    B = args.batch // world if args.scaling == 'strong' else args.batch
    elapsed, theta_end, ac = training_leg(d, T, B, args.steps, args.warmup)
    value = world * B * T * args.steps / elapsed
    del ac
    other = None
    c5 = None
    loops = None
    if multi:
        # both episode loops of the multi-GPU path in the same run: the headline above took the library's own RCCL loop
        # (mfg_train_rollouts_dist: all episodes of the call issued natively, ncclAllReduce on the launch stream) if its canary
        # passed on every rank, else the class's per-episode loop over torch.distributed; the other one is timed here
        from discrete_mean_field_game_amd import parallel
        native_on = bool(parallel.native_comm(None, dev, allow_single=True)) if args.backend == 'nccl' else False
        et, _, ac = training_leg(d, T, B, args.steps, args.warmup, native_rccl=False)
        del ac
        loops = {'headline_loop': 'native_rccl' if native_on else 'torch_dist',
                 'native_rccl_ms_per_update': (elapsed / args.steps * 1e3) if native_on else None,
                 'torch_dist_ms_per_update': et / args.steps * 1e3,
                 'torch_dist_value': world * B * T * args.steps / et,
                 'canary': (dict(parallel.CANARY_LOG[-1]) if parallel.CANARY_LOG else None)}
    if world > 1:
        # the other scaling mode, same run (not the headline value)
        Bo = args.batch if args.scaling == 'strong' else args.batch // world
        eo, _, ac = training_leg(d, T, Bo, args.steps, args.warmup)
        other = {'scaling': 'weak' if args.scaling == 'strong' else 'strong', 'batch_per_gpu': Bo,
                 'value': world * Bo * T * args.steps / eo, 'ms_per_step': eo / args.steps * 1e3, 'unit': 'env-steps/s'}
        del ac
        # BASELINE config 5: d=256, T=40, global batch 131072 sharded over the ranks
        B5 = max(1, args.c5_batch // world)
        e5, _, ac = training_leg(256, 40, B5, 2, 1)
        c5 = {'workload': 'C5 d=256 T=40 global batch %d split over %d GPUs' % (B5 * world, world), 'batch_per_gpu': B5,
              'value': world * B5 * 40 * 2 / e5, 'ms_per_step': e5 / 2 * 1e3, 'unit': 'env-steps/s', 'steps': 2}
        del ac

    collective = None
    if multi:
        # latency of the ONE exchange step of an update: all-reduce of the fused gradient buffer
        # [G_w | G_theta | sum r | count] (fp64, F+3 entries) and the replicated parameter update that follows it, timed
        # with events on the launch stream, every rank in lock-step
        Fc = ops.num_features(d)
        # (a zero buffer with count 1/world: the sums stay zero and the count entry stays 1 however often it is reduced)
        Gc = torch.zeros(Fc + 3, dtype=torch.float64, device=dev)
        wc = torch.zeros(Fc, dtype=torch.float64, device=dev)
        tc = torch.zeros(1, dtype=torch.float64, device=dev)

        def exchange(update):
            Gc[Fc + 2:].fill_(1.0 / world)                               # reset the count entry (it is summed too)
            all_reduce_(Gc)
            if update:
                ops.apply_update(Gc, d, 0.0, 0.0, wc, tc)
        sync()
        t_fill = event_time(lambda: Gc[Fc + 2:].fill_(1.0 / world), n=50, warm=10)
        sync()
        t_ar = event_time(lambda: exchange(False), n=50, warm=10) - t_fill
        sync()
        t_upd = event_time(lambda: exchange(True), n=50, warm=10) - t_fill
        sync()
        t_nat = None
        if args.backend == 'nccl':
            # the same exchange as the product issues it inside train(): ncclAllReduce from the HIP library's own communicator
            from discrete_mean_field_game_amd import parallel
            comm = parallel.native_comm(None, dev, allow_single=True)
            if comm:
                stream = torch.cuda.current_stream().cuda_stream
                t_nat = event_time(lambda: _lib.check(_lib.lib().mfg_dist_all_reduce(comm, Gc.data_ptr(), Fc + 3, stream), 'all_reduce'),
                                   n=50, warm=10)
                sync()
        collective = {'backend': ('rccl (torch.distributed nccl)' if args.backend == 'nccl'
                                  else 'gloo, staged through the host: wall time of a debug path, not a device collective'),
                      'world': world, 'payload_bytes': (Fc + 3) * 8, 'all_reduce_us': t_ar * 1e6,
                      'all_reduce_plus_apply_update_us': t_upd * 1e6,
                      'native_all_reduce_us': (t_nat * 1e6 if t_nat is not None else None),
                      'note': 'the headline leg exchanges G through the HIP library\'s own RCCL communicator (mfg_train_rollouts_dist, '
                              'native episode loop) when its canary passed on every rank (parallel.native_comm; `loops.canary`), '
                              'else through torch.distributed (one all-reduce per update, the update applied inside the next rollout '
                              'kernel); `loops` times both; native_all_reduce_us = ncclAllReduce of G issued by the library itself'}

    out = None
    if rank == 0:
        bytes_per_step = 4 * (d * d + 2 * d + 1)                         # SURVEY.md 8d / BASELINE.md 3
        roofline = None
        fused = None
        configs = None
        if not args.no_roofline:
            # the HBM-bound kernel of the path on THIS GPU: materialised actions of one rollout of the global batch
            # (B*T transitions, a 1.7 GB slab > L3 at the default size), the same leg at every N
            Br = args.batch if args.scaling == 'strong' else B
            gp = given_p_leg(d, Br, T)
            traffic, traffic_src = pmc_traffic('k_step_', d, T, Br)
            roofline = {'bound': 'hbm', 'achieved': gp['achieved_GBs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': gp['frac'], 'traffic': traffic, 'traffic_source': traffic_src,
                        'kernel': (('k_step_wave_batched' if gp['transitions'] >= 100000 else 'k_step_wave') if d in (21, 15) else 'k_step_small') if d <= 64 else ('k_step_rows' if d in (128, 256) else 'k_step_large'),
                        'leg': 'given-P transition+reward over %d transitions (P slab %.2f GB)' % (gp['transitions'], gp['slab_GB']),
                        'algorithmic_bytes_per_launch': gp['transitions'] * bytes_per_step,
                        'avg_launch_us': gp['avg_launch_us'], 'env_steps_per_s': gp['env_steps_per_s']}
            # this box's own streaming ceilings (stock torch kernels on a 1.6 GB buffer), for context
            xx = torch.empty(400_000_000, device=dev).uniform_()
            yy = torch.empty_like(xx)
            t_rd = event_time(lambda: xx.sum(), n=5)
            t_cp = event_time(lambda: yy.copy_(xx), n=5)
            roofline['box_ceiling_GBs'] = {'torch_sum_read': xx.numel() * 4 / t_rd / 1e9,
                                           'torch_copy_read_plus_write': 2 * xx.numel() * 4 / t_cp / 1e9}
            del xx, yy
            if world == 1:                                               # (a leg on rank 0 alone would hang the other ranks)
                F = ops.num_features(d)
                idx = torch.as_tensor(np.random.RandomState(1234).randint(64, size=B).astype(np.int32), device=dev)
                pi0 = ops.gather_start(start_table(d), idx)
                th = torch.tensor([theta0], dtype=torch.float64, device=dev)
                wv = torch.as_tensor(np.random.RandomState(1).rand(F), device=dev)
                Gf = torch.zeros(F + 3, dtype=torch.float64, device=dev)
                wsf = ops.workspace(B * T, d, dev)
                bf = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'pi_last': torch.empty(B, d, device=dev),
                      'reward': torch.empty(B, T, device=dev), 'delta': torch.empty(B, T, dtype=torch.float64, device=dev),
                      'g': torch.empty(B, T, dtype=torch.float64, device=dev)}
                t_f = event_time(lambda: ops.rollout(pi0, T, th, shift, alpha_scale, w=wv, gamma=gamma, seed=7, td=True, G=Gf,
                                                     ws=wsf, out=bf), n=10, warm=3)
                fused = {'kernel': ('k_core_small' if d <= 64 else 'k_core_large') + '<SAMPLE,TD,MIXED> + k_grad + k_reduce_partials',
                         'bound': 'valu/transcendental+rng (P stays on chip)', 'avg_launch_ms': t_f * 1e3,
                         'avg_launch_note': 'separate 10-launch probe of ops.rollout (rollout + batch sums, no update) after the timed '
                                            'region, not a share of ms_per_step: run-to-run scatter of ~2 % can put it above the headline',
                         'env_steps_per_s': B * T / t_f, 'hbm_algorithmic_GBs': B * T * bytes_per_step / t_f / 1e9}
                fused.update(pmc_sq('k_core_', d, T, B))
                # the kernel the HEADLINE times, on both lines: its share of the HBM roofline (it never sends P through HBM, so this
                # is small by construction) and of the VALU issue bound that actually limits it (committed SQ counters)
                fused['headline_frac_of_hbm_line'] = value * bytes_per_step / 1e9 / HBM_PEAK_GBS
                import glob as _glob
                _ct = sorted(_glob.glob(os.path.join(ROOT, 'profiles', 'r*_cycle_table_d%d.txt' % d)))
                fused['headline_bound'] = ('VALU issue: valu_busy %.2f of the kernel time (4 cycles booked per instruction: an upper bound), priced '
                                           'issue work ~0.75-0.9 of the SIMD cycles (%s); the roofline object above is the given-P kernel'
                                           % (fused.get('valu_busy', float('nan')),
                                              ('profiles/' + os.path.basename(_ct[-1])) if _ct else 'no cycle table committed for this d'))
                # host-boundness of the class API: the same updates issued by one native call, GPU time from events
                t_native = native_leg(d, T, B, args.steps, args.warmup)
                fused['native_loop_ms_per_step'] = t_native / args.steps * 1e3
                fused['class_api_gpu_busy_frac'] = min(1.0, t_native / elapsed)
                del pi0, idx, th, wv, Gf, wsf, bf
            if world == 1 and not args.no_configs:
                configs = other_configs(training_leg, native_leg, given_p_leg, d, T, B, args)
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(d, args.cpu_seconds)
        out = {
            'metric': 'env-steps/sec for batched d-bin population rollouts', 'value': value, 'unit': 'env-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32 (f64 accumulate)',
            'data': 'synthetic',
            'config': {'workload': 'actor_critic.train(update_every="rollout") of the drop-in class (mfg_ac2.train call surface; '
                                   'device-side start draw every episode, one update per 15-step rollout): '
                                   'd=%d topics, T=%d, global batch %d = %d trajectories per GPU x %d'
                                   % (d, T, B * world, B, world),
                       'd': d, 'T': T, 'batch_per_gpu': B, 'global_batch': B * world,
                       'theta0': theta0, 'shift': shift, 'alpha_scale': alpha_scale, 'rng': 'philox4x32-10',
                       'parallelism': 'trajectory-sharded x%d, 1 all-reduce/update' % world},
            'theta_end': theta_end,
            'modes': {'headline': "update_every='rollout' (one update per 15-step episode)",
                      'class_default': "update_every='step' (the reference's update after every env step, mfg_ac2.py:505-522): see the "
                                       "'update per env step' entries of `configs`"},
            'roofline': roofline, 'fused_kernel': fused, 'cpu_baseline': cpu,
        }
        if collective is not None:
            out['collective'] = collective
        if loops is not None:
            out['loops'] = loops
        if other is not None:
            out['other_scaling'] = other
        if c5 is not None:
            out['c5_strong'] = c5
        if configs is not None:
            out['configs'] = configs
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


def other_configs(training_leg, native_leg, given_p_leg, d0, T0, B0, args):
    """BASELINE.json's other configurations on this GPU, bounded (a few seconds each).  Every training number is a timed
    call of the drop-in class (actor_critic.train / AC_IRL.train); `class_api_gpu_busy_frac` = GPU time of the same updates
    issued by one native call (events) / wall time of the class call.  Plus the given-P kernel's HBM rate per shape and the
    headline shape in strict f64 precision."""
    import time as _t
    import numpy as np
    import torch
    out = []
    for name, d, T, B, steps, warm in (('C2', 21, 15, 4096, 60, 20), ('C3', 128, 40, 16384, 3, 1),
                                       ('C5 (1/8 share)', 256, 40, 16384, 2, 1)):
        try:
            e, _, st = training_leg(d, T, B, steps, warm)
            del st
            tn = native_leg(d, T, B, steps, warm)
            gp = given_p_leg(d, B, T if d <= 64 else 1, n_launch=20, warm=10)
            bps = 4 * (d * d + 2 * d + 1)
            out.append({'config': name + ': actor_critic.train(update_every="rollout"), class API', 'd': d, 'T': T, 'batch': B,
                        'steps': steps, 'fused_ms_per_rollout': e / steps * 1e3,
                        'fused_env_steps_per_s': B * T * steps / e, 'native_loop_ms_per_rollout': tn / steps * 1e3,
                        'class_api_gpu_busy_frac': min(1.0, tn / e),
                        'fused_frac_of_hbm_line': B * T * steps / e * bps / 1e9 / HBM_PEAK_GBS,
                        'given_P_GBs': gp['achieved_GBs'], 'given_P_frac': gp['frac'], 'given_P_slab_GB': gp['slab_GB'],
                        'given_P_avg_launch_us': gp['avg_launch_us']})
        except Exception as exc:  # informational: never fail the headline on it
            out.append({'config': name, 'error': repr(exc)})
    # strong-scaling shards of the headline batch on THIS GPU (what each rank of an N-GPU run executes per update before
    # its all-reduce: rollout + batch sums, then update), so the scaling projection is driver-visible without a node
    try:
        sh = []
        for n in (2, 4, 8):
            b = B0 // n
            if b < 1:
                continue
            e, _, st = training_leg(d0, T0, b, 40, 10)
            sh.append({'gpus': n, 'batch_per_gpu': b, 'ms_per_update': e / 40 * 1e3, 'env_steps_per_s_per_gpu': b * T0 * 40 / e})
            del st
        out.append({'config': 'strong-scaling shards of the headline batch through actor_critic.train, one GPU each (no collective)', 'd': d0, 'T': T0,
                    'shards': sh})
    except Exception as exc:
        out.append({'config': 'shards', 'error': repr(exc)})
    # round 6: the two lane mappings of the d = 21 sampling kernels at batches that under-fill the machine (mfg_set_core_mapping:
    # the packed k_core_small vs k_core_row3, one trajectory per wave; same bits) -- native loop, event timed, same box, same run
    try:
        from discrete_mean_field_game_amd import _lib as _L
        rows = []
        for Bs in (4096, 2048, 1024):
            e = {'batch': Bs}
            for mode, name in ((1, 'packed'), (2, 'row3')):
                _L.lib().mfg_set_core_mapping(mode)
                e[name + '_ms_per_update'] = native_leg(21, 15, Bs, 60, 20) / 60 * 1e3
            _L.lib().mfg_set_core_mapping(0)
            e['row3_over_packed'] = e['packed_ms_per_update'] / e['row3_ms_per_update']
            e['env_steps_per_s_row3'] = Bs * 15 / (e['row3_ms_per_update'] * 1e-3)
            rows.append(e)
        out.append({'config': 'd=21 lane mappings side by side: packed (3 trajectories per wave, a lane per matrix row) vs row3 (1 trajectory '
                              'per wave, 3 lanes per row; taken automatically up to 16 trajectories per CU); rollout + batch sums + update per episode',
                    'd': 21, 'T': 15, 'rows': rows})
    except Exception as exc:
        out.append({'config': 'lane mappings', 'error': repr(exc)})
    finally:
        try:
            _L.lib().mfg_set_core_mapping(0)
        except Exception:
            pass
    # reference semantics: theta and w move after EVERY env step (mfg_ac2.py:505-522), batch-mean gradient; the 15-step
    # episode is issued natively (mfg_train_episode: 15 x [fused step kernel | batch sums + update])
    try:
        for Bs, episodes in ((65536, 20), (4096, 100)):
            e, _, st = training_leg(21, 15, Bs, episodes, 3 if Bs > 4096 else 20, mode='step')
            del st
            out.append({'config': 'mfg_ac2.train, update per env step (reference semantics), class API', 'd': 21, 'T': 15,
                        'batch': Bs, 'episodes': episodes, 'env_steps_per_s': Bs * 15 * episodes / e,
                        'ms_per_episode': e / episodes * 1e3})
    except Exception as exc:
        out.append({'config': 'mfg_ac2.train step mode', 'error': repr(exc)})
    # headline shape, strict f64 policy math
    try:
        e, _, st = training_leg(d0, T0, B0, 5, 1, precision='f64')
        out.append({'config': 'headline shape, precision f64', 'd': d0, 'T': T0, 'batch': B0, 'steps': 5,
                    'fused_ms_per_rollout': e / 5 * 1e3, 'fused_env_steps_per_s': B0 * T0 * 5 / e})
        del st
    except Exception as exc:
        out.append({'config': 'headline f64', 'error': repr(exc)})
    # C4: max-ent IRL forward solve with the HIP reward network in the loop (ac_irl.py:634-732), d=21, B=4096
    try:
        from discrete_mean_field_game_amd.ac_irl import AC_IRL
        rs = np.random.RandomState(0)
        mat = rs.dirichlet(np.ones(21), size=64)
        # (20 untimed episodes first: after the few idle milliseconds between two legs the device takes ~5-10 ms of work to be back
        #  at its sustained clock -- tools/irl_rep_probe.py: 0.459 ms per episode right after an idle gap, 0.42 sustained)
        for mode, episodes in (('step', 100), ('rollout', 100)):
            np.random.seed(5); torch.manual_seed(5)
            ac = AC_IRL(theta=8.64, shift=0.0, alpha_scale=1e4, d=21, pi0=mat, demonstrations=[], batch=4096, seed=3,
                        update_every=mode, verbose=0)
            ac.train(max_episodes=20, stop_criteria=-1)
            torch.cuda.synchronize()
            t0 = _t.perf_counter()
            ac.train(max_episodes=episodes, stop_criteria=-1)
            torch.cuda.synchronize()
            dt = _t.perf_counter() - t0
            out.append({'config': 'C4 AC_IRL.train (reward net in the loop), update per %s' % mode, 'd': 21, 'T': 15,
                        'batch': 4096, 'episodes': episodes, 'env_steps_per_s': 4096 * 15 * episodes / dt,
                        'ms_per_episode': dt / episodes * 1e3,
                        'launches': ('2 per env step (step kernel carrying the previous step\'s row reduction | reward network + TD '
                                     'error + batch sums) + 1 row reduction per episode: 31 per episode' if mode == 'step' else
                                     '4 per episode (rollout | reward network | batch sums | row reduction + update)')})
    except Exception as exc:
        out.append({'config': 'C4', 'error': repr(exc)})
    # C4, the IRL experiment itself: AC_IRL.outerloop (ac_irl.py:900-954, what gridsearch.py:21-23 runs) = per iteration
    # [generate 5 trajectories into D_samp | reward_iteration(100 reward updates, eval every 10) | train(200 episodes)] with the
    # reference's defaults, 21 synthetic demonstrations; wall-time split (synchronised around each phase)
    try:
        import random as _random
        from discrete_mean_field_game_amd.ac_irl import AC_IRL
        rs = np.random.RandomState(0)
        mat = rs.dirichlet(np.ones(21), size=64)
        demos = [[(rs.dirichlet(np.ones(21)), rs.dirichlet(np.ones(21), size=21)) for _ in range(15)] for _ in range(21)]
        for mode in ('step', 'rollout'):
            np.random.seed(5); torch.manual_seed(5); _random.seed(5)
            ac = AC_IRL(theta=8.64, shift=0.0, alpha_scale=1e4, d=21, pi0=mat, demonstrations=demos, batch=4096, seed=3,
                        update_every=mode, verbose=0)
            split = {'generate': 0.0, 'reward_iteration': 0.0, 'train': 0.0}

            def timed(name, fn):
                def wrapped(*a, **k):
                    torch.cuda.synchronize(); t0 = _t.perf_counter()
                    r = fn(*a, **k)
                    torch.cuda.synchronize(); split[name] += _t.perf_counter() - t0
                    return r
                return wrapped
            ac._generate_device = timed('generate', ac._generate_device)
            ac.reward_iteration = timed('reward_iteration', ac.reward_iteration)
            ac.train = timed('train', ac.train)
            ac.outerloop(num_iterations=1, final_training=False)                    # warm-up iteration (untimed)
            for k in split:
                split[k] = 0.0
            iters = 2
            torch.cuda.synchronize(); t0 = _t.perf_counter()
            ac.outerloop(num_iterations=iters, final_training=False)
            torch.cuda.synchronize(); dt = _t.perf_counter() - t0
            n_upd = ac.reward_update_count / iters
            out.append({'config': 'C4 AC_IRL.outerloop (reward learning + forward solve on the device), update per %s' % mode,
                        'd': 21, 'T': 15, 'batch': 4096, 'outer_iterations': iters, 'ms_per_outer_iteration': dt / iters * 1e3,
                        'split_ms': {k: v / iters * 1e3 for k, v in split.items()},
                        'reward_updates_per_iteration': n_upd, 'us_per_reward_update': split['reward_iteration'] / iters / max(n_upd, 1) * 1e6,
                        'forward_episodes_per_iteration': 200,
                        'env_steps_per_s_whole_loop': 4096 * 15 * 200 * iters / dt,
                        'reward_iteration_share': split['reward_iteration'] / max(dt, 1e-12),
                        'loss': ac.loss_val, 'theta_end': float(np.ravel(ac.theta)[0])})
    except Exception as exc:
        out.append({'config': 'C4 outerloop', 'error': repr(exc)})
    return out


if __name__ == '__main__':
    main()
