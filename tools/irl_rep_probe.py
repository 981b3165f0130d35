"""Developer probe (round 6): AC_IRL.train in step mode, several timed calls back to back and one after an idle gap -- wall time against
GPU event time per episode (is the class call host bound? how much does a short call after idle time lose to the clock ramp?)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from discrete_mean_field_game_amd.ac_irl import AC_IRL
rs = np.random.RandomState(0)
mat = rs.dirichlet(np.ones(21), size=64)
np.random.seed(5); torch.manual_seed(5)
ac = AC_IRL(theta=8.64, shift=0.0, alpha_scale=1e4, d=21, pi0=mat, demonstrations=[], batch=4096, seed=3, update_every='step', verbose=0)
ac.train(max_episodes=3, stop_criteria=-1)
torch.cuda.synchronize()
for rep in range(6):
    n = 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    ac.train(max_episodes=n, stop_criteria=-1)
    e1.record(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('rep %d: wall %.4f ms per episode, GPU events %.4f ms per episode' % (rep, dt / n * 1e3, e0.elapsed_time(e1) / n))
    if rep == 2:
        time.sleep(0.5)     # let the device idle
# host issue rate: how long does the host need to ISSUE an episode (31 launches)?  (queue kept short by syncing first)
from discrete_mean_field_game_amd import ops
t0 = time.perf_counter()
ac.train(max_episodes=200, stop_criteria=-1)
t1 = time.perf_counter()
torch.cuda.synchronize()
print('200 episodes: wall until train() returned %.4f ms per episode' % ((t1 - t0) / 200 * 1e3))
