"""rocprofv3 --pmc driver: a few launches of the fused training rollout kernel at the bench shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops
d, T, B = 21, 15, 65536
if len(sys.argv) > 1:
    d, T, B = (int(x) for x in sys.argv[1].split(','))
dev = torch.device('cuda:0')
if os.environ.get('MFG_MAPPING'):   # 1 / 2: force the packed / the one-trajectory-per-wave lane mapping (A/B counter runs)
    from discrete_mean_field_game_amd import _lib
    _lib.lib().mfg_set_core_mapping(int(os.environ['MFG_MAPPING']))
th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
rs = np.random.RandomState(0)
pi0 = torch.as_tensor(rs.dirichlet(np.ones(d), size=B).astype(np.float32), device=dev)
w = torch.as_tensor(rs.rand(ops.num_features(d)), device=dev)
td = 'env' not in sys.argv[2:]
for _ in range(int(os.environ.get('PMC_LAUNCHES', '3'))):
    ops.rollout(pi0, T, th, 0.16, 12000.0, w=w if td else None, seed=7, td=td)
torch.cuda.synchronize()
