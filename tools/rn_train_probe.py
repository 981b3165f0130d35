"""Event-timed reward-network training step (mfg_reward_net_train_step: 2 launches) at the C4 shape, back to back without the
class around it, and the same through AC_IRL.update_reward (host cost of the Python call path).  Usage: python tools/rn_train_probe.py [d]"""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discrete_mean_field_game_amd.ac_irl import AC_IRL  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 21
dev = torch.device('cuda:0')
rs = np.random.RandomState(0)
mat = rs.dirichlet(np.ones(d), size=64)
demos = [[(rs.dirichlet(np.ones(d)), rs.dirichlet(np.ones(d), size=d)) for _ in range(15)] for _ in range(21)]
ac = AC_IRL(theta=8.64, shift=0, alpha_scale=1e4, d=d, pi0=mat, demonstrations=demos, batch=64, seed=3, verbose=0, device=dev)
ac._gen_store.push(*ac._generate_device(50))
tr = ac._trainer
rows_d = [ac._demo_store.rows[i] for i in range(5)]
rows_g = [ac._gen_store.rows[i] for i in range(5)]
for _ in range(20):
    tr.step(ac._demo_store, rows_d, ac._gen_store, rows_g, 5, 1)
torch.cuda.synchronize()
n = 300
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for k in range(n):
    tr.step(ac._demo_store, rows_d, ac._gen_store, rows_g, 5, k)
e1.record()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
print('RewardTrainer.step d=%d: %.1f us per update on the stream (events), host issue %.1f us per call' % (d, e0.elapsed_time(e1) * 1e3 / n, t_issue * 1e6 / n))
random.seed(0)
t0 = time.perf_counter()
for k in range(n):
    ac.update_reward()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('AC_IRL.update_reward: host issue %.1f us per call, wall %.1f us per update' % (t_issue * 1e6 / n, t_all * 1e6 / n))
