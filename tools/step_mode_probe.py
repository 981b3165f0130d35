"""Driver for tools/trace_gaps.sh: a few episodes of mfg_ac2.train with per-step updates at a small batch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd.mfg_ac2 import actor_critic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
np.random.seed(0)
ac = actor_critic(d=21, batch=B, update_every='step', verbose=0)
ac.train(6, consecutive=1000)
torch.cuda.synchronize()
