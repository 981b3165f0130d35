"""Wall-time split of the IRL experiment's outer loop (reference ac_irl.py:900-954: generate | reward_iteration | train) at the
C4 shape, HIP reward learning vs the PyTorch-autograd update (MODE=torch: the round-4 path).  Usage:
    python tools/irl_outer_probe.py [B] [iterations] [reward_iterations] [episodes]"""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discrete_mean_field_game_amd.ac_irl import AC_IRL  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
RIT = int(sys.argv[3]) if len(sys.argv) > 3 else 100
EPS = int(sys.argv[4]) if len(sys.argv) > 4 else 200
d = 21
dev = torch.device('cuda:0')


def run(mode, update_every):
    rs = np.random.RandomState(0)
    mat = rs.dirichlet(np.ones(d), size=64)
    demos = [[(rs.dirichlet(np.ones(d)), rs.dirichlet(np.ones(d), size=d)) for _ in range(15)] for _ in range(21)]
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    ac = AC_IRL(theta=8.64, shift=0, alpha_scale=1e4, d=d, pi0=mat, demonstrations=demos, batch=B, seed=3, verbose=0,
                update_every=update_every, device=dev)
    if mode == 'torch':
        ac._trainer = None
    split = {'generate': 0.0, 'reward_iteration': 0.0, 'train': 0.0}

    def timed(name, fn):
        def w(*a, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = fn(*a, **k)
            torch.cuda.synchronize(); split[name] += time.perf_counter() - t0
            return out
        return w
    ac._generate_device = timed('generate', ac._generate_device)
    ac.generate_trajectories = timed('generate', ac.generate_trajectories)
    ac.reward_iteration = timed('reward_iteration', ac.reward_iteration)
    ac.train = timed('train', ac.train)
    ac.outerloop(num_iterations=1, max_reward_iterations=RIT, max_forward_episodes=EPS, final_training=False)   # warm-up
    for k in split:
        split[k] = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ac.outerloop(num_iterations=ITERS, max_reward_iterations=RIT, max_forward_episodes=EPS, final_training=False)
    torch.cuda.synchronize(); total = time.perf_counter() - t0
    per = {k: 1e3 * v / ITERS for k, v in split.items()}
    print('%-6s update_every=%-7s B=%d: outer iteration %.1f ms = generate %.2f | reward_iteration(%d) %.2f (%.1f us/update) | train(%d) %.2f'
          '   [loss %.4f theta %.5f]' % (mode, update_every, B, 1e3 * total / ITERS, per['generate'], RIT, per['reward_iteration'],
                                         1e3 * per['reward_iteration'] / RIT, EPS, per['train'], ac.loss_val, float(np.ravel(ac.theta)[0])))


for ue in ('step', 'rollout'):
    for mode in ('hip', 'torch'):
        run(mode, ue)
