"""Randomised soak of the IRL env step issued natively (mfg_train_episode_irl_draw: two launches per env step where the matrix-core
reward-network kernel serves -- step kernel that forms theta from the previous step's partial rows and carries their reduction,
network launch that forms the TD error -- three otherwise) against the per-step Python sequence of separate kernels (step kernel
with its own value part | plain network forward | gradient kernel + update), through the drop-in class (developer tool).
Random d in {21, 15, 12}, batch 1 .. 6000, n_fc3 1 .. 24, n_fc4 1 .. 32, regulariser variant, precision, gamma, learning rates,
episodes 1 .. 3.  Reports the largest relative deviation of theta and w.   usage: irl_step_soak.py [seconds]"""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd.ac_irl import AC_IRL
from discrete_mean_field_game_amd.networks import REG_VARIANTS
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(20261005)
t0 = time.time()
n = 0
worst = {}
while time.time() - t0 < budget:
    d = int(rs.choice([21, 21, 21, 15, 15, 12]))
    B = int(rs.choice([rs.randint(1, 40), rs.randint(40, 700), rs.randint(700, 6000)]))
    n3 = int(rs.choice([1, 5, 8, 8, 8, 16, 24]))
    n4 = int(rs.choice([1, 4, 4, 8, 17, 32]))
    reg = REG_VARIANTS[rs.randint(4)]
    precision = 'mixed' if rs.rand() < 0.7 else 'f64'
    gamma = float(rs.choice([1.0, 0.95, 0.6]))
    episodes = int(rs.randint(1, 4))
    seed = int(rs.randint(1 << 30))
    mat = rs.dirichlet(np.ones(d) * rs.choice([0.5, 1.0, 3.0]), size=int(rs.randint(1, 40)))
    lr_c, lr_a = float(rs.choice([0.1, 0.01])), float(rs.choice([1e-3, 1e-4]))
    runs = []
    for native in (True, False):
        np.random.seed(seed & 0xFFFF); torch.manual_seed(seed); random.seed(seed)
        ac = AC_IRL(theta=float(8.0 + (seed % 100) / 100.0), shift=0.1, d=d, pi0=mat, demonstrations=[], batch=B, rng='philox',
                    seed=seed % 1000, update_every='step', precision=precision, reg=reg, n_fc3=n3, n_fc4=n4, verbose=0)
        with torch.no_grad():
            for p in ac.reward_net.parameters():
                if p.dim() == 1:
                    p.uniform_(-0.2, 0.2)
        if not native:
            ac.trace = []                       # tracing forces the per-step Python path
        ac.train(max_episodes=episodes, stop_criteria=-1, gamma=gamma, lr_critic=lr_c, lr_actor=lr_a, constant=bool(seed & 1),
                 consecutive=1000)
        runs.append((float(np.ravel(ac.theta)[0]), ac.w.copy()))
    fam = 'd=%d %s' % (d, 'two launches (n_fc3 <= 16)' if (d in (21, 15) and n3 <= 16) else 'three launches')
    dth = abs(runs[0][0] - runs[1][0]) / abs(runs[1][0])
    dw = float(np.max(np.abs(runs[0][1] - runs[1][1])) / max(np.max(np.abs(runs[1][1])), 1e-300))
    wv = worst.setdefault(fam, [0.0, 0.0, 0, None])
    wv[2] += 1
    if max(dth, dw) > max(wv[0], wv[1]):
        wv[3] = dict(d=d, B=B, n3=n3, n4=n4, reg=reg, precision=precision, gamma=gamma, episodes=episodes, seed=seed)
    wv[0], wv[1] = max(wv[0], dth), max(wv[1], dw)
    if not (np.isfinite(runs[0][0]) and np.all(np.isfinite(runs[0][1]))):
        print('NON-FINITE', fam, dict(d=d, B=B, n3=n3, n4=n4, reg=reg, precision=precision, seed=seed)); sys.exit(1)
    n += 1
print('%d random configurations in %.0f s' % (n, time.time() - t0))
bad = False
for fam, (dth, dw, cnt, cfg) in sorted(worst.items()):
    print('%-40s %5d cases: largest relative deviation theta %.2e, w %.2e   (worst: %s)' % (fam, cnt, dth, dw, cfg))
    bad = bad or dth > 1e-10 or dw > 1e-10
print('FAIL' if bad else 'OK (bar: 1e-10 -- the two sequences associate the fp64 batch sums differently)')
sys.exit(1 if bad else 0)
