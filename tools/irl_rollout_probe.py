"""Driver for tools/trace_gaps.sh: a few AC_IRL.train episodes in rollout mode (one update per episode) at batch B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd.ac_irl import AC_IRL
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 21
rs = np.random.RandomState(0)
irl = AC_IRL(theta=8.64, d=d, batch=B, demonstrations=[], pi0=rs.dirichlet(np.ones(d), size=16), verbose=0, update_every='rollout')
irl.train(8, stop_criteria=-1, consecutive=1000)
torch.cuda.synchronize()
