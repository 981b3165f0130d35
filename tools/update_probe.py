"""Event-timed full training updates (mfg_train_rollouts: device draw + rollout + batch sums + update), back to back.
usage: update_probe.py [d,T,B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops
dev = torch.device('cuda:0')
cfgs = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]] or [(21, 15, 65536), (21, 15, 8192), (21, 15, 4096)]
for d, T, B in cfgs:
    rs = np.random.RandomState(0)
    mat = torch.as_tensor(rs.dirichlet(np.ones(d), size=64).astype(np.float32), device=dev)
    F = ops.num_features(d)
    w = torch.as_tensor(rs.rand(F), device=dev)
    th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
    G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
    ws = ops.workspace(B * T, d, dev)
    bufs = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'pi_last': torch.empty(B, d, device=dev), 'reward': torch.empty(B, T, device=dev),
            'delta': torch.empty(B, T, dtype=torch.float64, device=dev), 'g': torch.empty(B, T, dtype=torch.float64, device=dev)}
    n = 40 if d <= 64 else 3
    run = lambda k, e0: ops.train_rollouts(mat, T, k, e0, False, th, 0.16, 12000.0, w, 1.0, G, ws, bufs, 0.1, 0.001, seed=1, first_step=e0 * T)
    run(10 if d <= 64 else 1, 0)
    best = 1e9
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); run(n, 10 + rep * n); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    print('d=%d T=%d B=%6d: %.4f ms per update  %.3e env-steps/s  theta %.12f' % (d, T, B, best, B * T / best * 1e3, float(th[0])), flush=True)
