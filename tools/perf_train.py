"""Throughput of the drop-in classes' train() loops (configs C2 / C4 of BASELINE.json): env-steps/s incl. host overhead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd.mfg_ac2 import actor_critic
from discrete_mean_field_game_amd.ac_irl import AC_IRL

def run(name, ac, episodes, **kw):
    ac.train(2, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); ac.train(episodes, **kw); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print('%-46s %7.2f ms/episode  %.3e env-steps/s' % (name, t / episodes * 1e3, ac.batch * 15 * episodes / t))

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
np.random.seed(0)
for mode in ('step', 'rollout'):
    ac = actor_critic(d=21, batch=B, update_every=mode, verbose=0)
    run('mfg_ac2 d=21 B=%d update_every=%s' % (B, mode), ac, 30, consecutive=1000)
rs = np.random.RandomState(0)
d = 21
demos = [[(rs.dirichlet(np.ones(d)), rs.dirichlet(np.ones(d), size=d)) for _ in range(15)] for _ in range(6)]
irl = AC_IRL(theta=8.64, d=d, batch=B, demonstrations=demos, verbose=0)
run('ac_irl d=21 B=%d (reward net per step)' % B, irl, 10, stop_criteria=-1, consecutive=1000)
irl2 = AC_IRL(theta=8.64, d=d, batch=B, demonstrations=demos, update_every='rollout', verbose=0)
run('ac_irl d=21 B=%d (one update per episode)' % B, irl2, 10, stop_criteria=-1, consecutive=1000)
