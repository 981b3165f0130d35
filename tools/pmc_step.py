"""Tiny driver for rocprofv3 --pmc passes: a few launches of the HBM-bound given-P kernel (and the fused
rollout kernel) at the bench's default shape, nothing else."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops

d, T, B = 21, 15, 65536
if len(sys.argv) > 1:
    d, T, B = (int(x) for x in sys.argv[1].split(','))
dev = torch.device('cuda:0')
th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
rs = np.random.RandomState(0)
pi0 = torch.as_tensor(rs.dirichlet(np.ones(d), size=B).astype(np.float32), device=dev)
w = torch.as_tensor(rs.rand(ops.num_features(d)), device=dev)
r = ops.rollout(pi0, T, th, 0.16, 12000.0, w=w, seed=7, td=True, write_P=True)
N = B * T
P_all = r['P'].view(N, d, d)
pi_all = r['pi_traj'][:, :T].contiguous().view(N, d)
for _ in range(5):
    ops.step_given_P(pi_all, P_all)
torch.cuda.synchronize()
print('done', N, 'transitions; algorithmic bytes per step_given_P launch =', N * 4 * (d * d + 2 * d + 1))
