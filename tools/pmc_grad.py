"""rocprofv3 --pmc driver: a few launches of the small-d batch-sum kernel on the outputs of a bench-shape rollout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops
d, T, B = 21, 15, 65536
dev = torch.device('cuda:0')
th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
rs = np.random.RandomState(0)
pi0 = torch.as_tensor(rs.dirichlet(np.ones(d), size=B).astype(np.float32), device=dev)
F = ops.num_features(d)
w = torch.as_tensor(rs.rand(F), device=dev)
out = ops.rollout(pi0, T, th, 0.16, 12000.0, w=w, seed=7, td=True)
G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
ws = ops.workspace(B * T, d, dev)
for _ in range(int(os.environ.get('PMC_LAUNCHES', '5'))):
    ops.grad_accumulate(out['pi_traj'], out['delta'].view(-1), out['g'].view(-1), out['reward'].view(-1), G, ws, T=T)
torch.cuda.synchronize()
