"""mfg_ac2.train with per-step updates at B = 65 536 (developer tool): ms per episode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd.mfg_ac2 import actor_critic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
np.random.seed(0)
ac = actor_critic(d=21, batch=B, update_every='step', verbose=0)
ac.train(5, consecutive=10 ** 9); torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter(); ac.train(20, consecutive=10 ** 9, first_episode=5); torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20)
print('mfg_ac2 step mode B=%d: %.3f ms/episode  %.3e env-steps/s' % (B, best * 1e3, B * 15 / best))
