"""C4 step mode through the class (AC_IRL.train, reward net per env step): ms per 15-step episode.  usage: irl_step_probe.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd.ac_irl import AC_IRL
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rs = np.random.RandomState(0)
mat = rs.dirichlet(np.ones(21), size=64)
for mode, episodes in (('step', 40), ('rollout', 60)):
    np.random.seed(5); torch.manual_seed(5)
    ac = AC_IRL(theta=8.64, shift=0.0, alpha_scale=1e4, d=21, pi0=mat, demonstrations=[], batch=B, seed=3, update_every=mode, verbose=0)
    ac.train(max_episodes=5, stop_criteria=-1)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        ac.train(max_episodes=episodes, stop_criteria=-1)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / episodes)
    print('AC_IRL.train B=%d update per %s: %.4f ms per episode  %.3e env-steps/s' % (B, mode, best * 1e3, B * 15 / best), flush=True)
