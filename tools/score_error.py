"""Developer check: relative error of the mixed-precision score g of a fused rollout against the strict fp64 score on
the SAME sampled actions (mfg_score, precision f64).  usage: score_error.py [d,B,T ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from discrete_mean_field_game_amd import ops

dev = torch.device('cuda:0')
cfgs = [(21, 4096, 15), (15, 4096, 15), (128, 256, 5), (256, 64, 3)]
if len(sys.argv) > 1:
    cfgs = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
for d, B, T in cfgs:
    rs = np.random.RandomState(1)
    pi0 = torch.as_tensor(rs.dirichlet(np.ones(d), size=B).astype(np.float32), device=dev)
    th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
    w = torch.as_tensor(rs.rand(ops.num_features(d)), device=dev)
    o = ops.rollout(pi0, T, th, 0.16, 12000.0, w=w, seed=3, td=True, write_P=True, precision='mixed')
    g = o['g'].cpu().numpy()
    errs = []
    for t in range(T):
        g64 = ops.score(o['pi_traj'][:, t].contiguous(), o['P'][:, t].contiguous(), th, 0.16, precision='f64').cpu().numpy()
        errs.append(np.max(np.abs(g[:, t] - g64) / np.maximum(np.abs(g64), 1e-30)))
    print('d=%d B=%d T=%d: max relative error of the mixed score vs strict fp64 on the same actions: %.2e (median |g| %.3g)'
          % (d, B, T, max(errs), np.median(np.abs(g))), flush=True)
