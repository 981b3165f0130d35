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
    print('d=%d T=%d B=%6d  rollout kernel %8.1f us   rollout+sums %8.1f us   full update %8.1f us   %.3e env-steps/s'
          % (d, T, B, t_core * 1e6, t_sums * 1e6, t_upd * 1e6, B * T / t_upd), flush=True)
    return t_core, t_sums, t_upd


if __name__ == '__main__':
    a = [int(x) for x in sys.argv[1:]]
    d = a[0] if a else 21
    T = a[1] if len(a) > 1 else 15
    Bs = a[2:] if len(a) > 2 else [65536, 32768, 16384, 8192, 4096, 2048, 1024]
    res = {B: probe(d, T, B) for B in Bs}
    if 65536 in res and 8192 in res:
        for ar in (0.0, 20e-6, 30e-6):
            print('projected strong scaling, all-reduce + apply = %2.0f us: ' % (ar * 1e6)
                  + '  '.join('%dx: %.2f' % (n, res[65536][2] / (res[65536 // n][1] + ar)) for n in (2, 4, 8) if 65536 // n in res))
