"""Per-update time of the forward-RL training rollout at the shard sizes of a strong-scaling run (developer tool / the
source of profiles/rNN_shards.txt): the fused rollout kernel alone, the whole single-GPU update (mfg_train_rollout with
MFG_TRAIN_APPLY: rollout + batch sums + parameter update), and the update without the apply (what a rank of a multi-GPU
job runs before its all-reduce).  Event timed on the launch stream, back-to-back launches.
usage: shard_table.py [d] [T] [B ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops

dev = torch.device('cuda:0')
# MFG_MAPPING = 1 / 2: force the packed / the one-trajectory-per-wave lane mapping of the d = 21 kernels (mfg_set_core_mapping) for A/B runs
if os.environ.get('MFG_MAPPING'):
    from discrete_mean_field_game_amd import _lib
    _lib.lib().mfg_set_core_mapping(int(os.environ['MFG_MAPPING']))
    print('# lane mapping forced to mode %s' % os.environ['MFG_MAPPING'], flush=True)


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e-3)
    return best


def probe(d, T, B, prec='mixed'):
    th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
    rs = np.random.RandomState(0)
    mat = torch.as_tensor(rs.dirichlet(np.ones(d), size=64).astype(np.float32), device=dev)
    idx = torch.as_tensor(rs.randint(64, size=B).astype(np.int32), device=dev)
    pi = ops.gather_start(mat, idx)
    F = ops.num_features(d)
    w = torch.as_tensor(rs.rand(F), device=dev)
    G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
    ws = ops.workspace(B * T, d, dev)
    bufs = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'reward': torch.empty(B, T, device=dev),
            'delta': torch.empty(B, T, dtype=torch.float64, device=dev), 'g': torch.empty(B, T, dtype=torch.float64, device=dev)}
    out = dict(bufs, pi_last=torch.empty(B, d, device=dev))
    # rollout kernel alone (TD outputs, no batch sums): mfg_rollout with reward_kind EXTERNAL computes everything but needs no G
    t_core = timeit(lambda: ops.rollout(pi, T, th, 0.16, 12000.0, w=w, seed=1, td=True, reward_kind=2, out=out, precision=prec))
    t_sums = timeit(lambda: ops.train_rollout(mat, idx, T, th, 0.16, 12000.0, w, 1.0, G, ws, bufs, 0.0, 0.0, apply=False, seed=1, precision=prec))
    t_upd = timeit(lambda: ops.train_rollout(mat, idx, T, th, 0.16, 12000.0, w, 1.0, G, ws, bufs, 0.0, 0.0, apply=True, seed=1, precision=prec))
    # a rank's cycle in a multi-GPU job (round 4): rollout with the PREVIOUS update applied in its weight staging | sums --
    # no update launch; start rows drawn in the kernel.  (The all-reduce between two cycles is not part of this number.)
    th2, w2 = torch.empty_like(th), torch.empty_like(w)
    state = {'p': (th, w), 'q': (th2, w2)}

    def cycle():
        (ta, wa), (tb, wb) = state['p'], state['q']
        ops.train_rollout_deferred(mat, None, T, ta, wa, (G, 0.0, 0.0, None), tb, wb, 0.16, 12000.0, 1.0, G, ws, bufs, seed=1,
                                   precision=prec)
        state['p'], state['q'] = (tb, wb), (ta, wa)
    G[F + 2] = 1.0
    t_cyc = timeit(cycle)
    print('d=%d T=%d B=%6d  rollout kernel %8.1f us   rollout+sums %8.1f us   full update %8.1f us   multi-rank cycle (deferred update) %8.1f us   %.3e env-steps/s'
          % (d, T, B, t_core * 1e6, t_sums * 1e6, t_upd * 1e6, t_cyc * 1e6, B * T / t_upd), flush=True)
    return t_core, t_sums, t_upd, t_cyc


if __name__ == '__main__':
    a = [int(x) for x in sys.argv[1:]]
    d = a[0] if a else 21
    T = a[1] if len(a) > 1 else 15
    Bs = a[2:] if len(a) > 2 else [65536, 32768, 16384, 8192, 4096, 2048, 1024]
    res = {B: probe(d, T, B) for B in Bs}
    if 65536 in res and 8192 in res:
        # exchange step = ONE all-reduce of G (the update itself rides in the next rollout).  Measured input: the 1-rank RCCL
        # all-reduce of the 2 KB buffer on this box (profiles/rNN_collective_1rank.json: ~10 us of launch + kernel); an
        # 8-rank ring over xGMI adds hops, so 10 us is the floor, 20 / 30 us the assumption of earlier rounds
        ar_meas = None
        try:
            import json, glob
            f = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r*_collective_1rank.json')))
            if f:
                ar_meas = json.load(open(f[-1]))['collective']['all_reduce_us'] * 1e-6
        except Exception:
            pass
        for ar in [0.0] + ([ar_meas] if ar_meas else []) + [20e-6, 30e-6]:
            print('projected strong scaling, all-reduce = %4.1f us%s: ' % (ar * 1e6, ' (measured, 1 rank)' if ar is ar_meas and ar_meas else '')
                  + '  '.join('%dx: %.2f' % (n, res[65536][2] / (res[65536 // n][3] + ar)) for n in (2, 4, 8) if 65536 // n in res))
