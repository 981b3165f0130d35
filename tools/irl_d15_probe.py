"""Developer probe (round 6): AC_IRL.train at the reference's default size d = 15 (ac_irl.py:33), both update modes, both lane mappings.
usage: irl_d15_probe.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import _lib as L
from discrete_mean_field_game_amd.ac_irl import AC_IRL
d = 15
rs = np.random.RandomState(0)
mat = rs.dirichlet(np.ones(d), size=64)
for B in ([int(x) for x in sys.argv[1:]] or [4096, 1024]):
    for mode, episodes in (('step', 60), ('rollout', 100)):
        row = 'AC_IRL.train d=15 B=%d update per %s:' % (B, mode)
        for m, name in ((1, 'packed'), (2, 'row4')):
            L.lib().mfg_set_core_mapping(m)
            np.random.seed(5); torch.manual_seed(5)
            ac = AC_IRL(theta=8.64, shift=0.0, alpha_scale=1e4, d=d, pi0=mat, demonstrations=[], batch=B, seed=3, update_every=mode, verbose=0)
            ac.train(max_episodes=20, stop_criteria=-1)
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                ac.train(max_episodes=episodes, stop_criteria=-1)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / episodes)
            row += '  %s %.4f ms per episode (%.3e env-steps/s)' % (name, best * 1e3, B * 15 / best)
        print(row, flush=True)
L.lib().mfg_set_core_mapping(0)
