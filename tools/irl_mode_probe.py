"""Driver for tools/trace_gaps.sh: a few AC_IRL.train episodes (reward net per step) at batch B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd.ac_irl import AC_IRL
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 21
rs = np.random.RandomState(0)
demos = [[(rs.dirichlet(np.ones(d)), rs.dirichlet(np.ones(d), size=d)) for _ in range(15)] for _ in range(6)]
irl = AC_IRL(theta=8.64, d=d, batch=B, demonstrations=demos, pi0=rs.dirichlet(np.ones(d), size=16), verbose=0)
irl.train(5, stop_criteria=-1, consecutive=1000)
torch.cuda.synchronize()
