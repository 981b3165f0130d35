"""Randomised soak of the reward-network kernels against the NumPy oracle (developer tool; uses oracle/ as the checker only).
Random (d in {21, 15, other}, n_fc3, n_fc4, B, dropout seed / offset); reports the largest deviation per kernel family.
usage: rn_soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops
from discrete_mean_field_game_amd.networks import RewardNet
from oracle import reward_net_oracle as RO
dev = torch.device('cuda:0')
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(20261003)
worst = {}
n = 0
t0 = time.time()
while time.time() - t0 < budget:
    d = int(rs.choice([21, 21, 21, 15, 15, 9, 28]))
    n3 = int(rs.choice([1, 2, 5, 8, 8, 8, 11, 16, 17, 24, 32]))
    n4 = int(rs.choice([1, 3, 4, 4, 4, 7, 8, 16, 32]))
    B = int(rs.choice([1, 2, 15, 16, 17, 100, 333, 1000, 4096, 4111, 5000]))
    drop = bool(rs.randint(2))
    torch.manual_seed(int(rs.randint(1 << 30)))
    net = RewardNet(d=d, reg='dropout_l1l2' if drop else 'none', n_fc3=n3, n_fc4=n4).to(dev).eval()
    for p in net.parameters():
        if p.dim() == 1:
            torch.nn.init.uniform_(p, -0.2, 0.2)
    state = rs.dirichlet(np.ones(d), size=B).astype(np.float32)
    action = rs.dirichlet(np.ones(d) * rs.choice([0.2, 1.0, 5.0]), size=(B, d)).astype(np.float32)
    s_t, a_t = torch.as_tensor(state, device=dev), torch.as_tensor(action, device=dev)
    params = RO.params_from_torch(net)
    if drop:
        seed, off = int(rs.randint(1 << 62)), int(rs.choice([0, 77, (1 << 33) + 5]))
        out = ops.reward_net_forward(net, s_t, a_t, seed=seed, sample_offset=off).cpu().numpy()
        ref = RO.forward(params, state.astype(np.float64), action.astype(np.float64), dropout=(float(net.keep_prob), seed, off))[:, 0]
    else:
        out = ops.reward_net_forward(net, s_t, a_t, dropout=False).cpu().numpy()
        ref = RO.forward(params, state.astype(np.float64), action.astype(np.float64))[:, 0]
    err = float(np.max(np.abs(out - ref)))
    fam = 'matrix-core (d=%d, n3<=16)' % d if d in (21, 15) and n3 <= 16 else ('run-mapped (d=%d, n3>16)' % d if d in (21, 15) else 'generic (d=%d)' % d)
    w = worst.setdefault(fam, [0.0, None, 0])
    w[2] += 1
    if err > w[0]:
        w[0], w[1] = err, (d, n3, n4, B, drop)
    assert err < 1e-5, (err, d, n3, n4, B, drop)
    n += 1
print('%d random configurations in %.0f s, every sample within 1e-5 of the fp64 oracle' % (n, time.time() - t0))
for fam in sorted(worst):
    print('  %-28s %4d configurations, largest |kernel - oracle| %.2e at (d, n3, n4, B, dropout) = %s' % (fam, worst[fam][2], worst[fam][0], worst[fam][1]))
