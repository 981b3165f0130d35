"""Randomised soak of the reward-network TRAINING step (mfg_reward_net_train_step) against the fp64 analytic gradient and the
tf.train.AdamOptimizer formula of oracle/reward_net_oracle.py (developer tool; uses oracle/ as the checker only).
Random geometry (d, conv sizes, filters, n_fc3, n_fc4), batch composition (1 .. 12 trajectories per half, permuted store rows),
regulariser variant and dropout key; three consecutive updates per configuration.  Reports, per kernel template, the largest
gradient deviation relative to the magnitude of what was summed (see tests/test_gpu_reward_train.py::_grad_scale) and the largest
parameter deviation after the Adam steps in units of the learning rate.   usage: rn_train_soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd.networks import RewardNet, REG_VARIANTS
from discrete_mean_field_game_amd.reward_learning import RewardTrainer, TrajectoryStore
from oracle import reward_net_oracle as RO
dev = torch.device('cuda:0')
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(20261004)
worst = {}
n = 0
skipped = 0
t0 = time.time()


def batch_np(store, logical):
    s, a = store.gather(logical)
    return s.reshape(-1, store.d).cpu().numpy().astype(np.float64), a.reshape(-1, store.d, store.d).cpu().numpy().astype(np.float64)


while time.time() - t0 < budget:
    ref_geom = rs.rand() < 0.6
    d = int(rs.choice([21, 21, 15, 15, 4, 9, 12, 28, 32]))
    k1, f2, k2 = (5, 2, 3) if ref_geom else (int(rs.choice([1, 3, 5, 7])), int(rs.choice([1, 2])), int(rs.choice([1, 3, 5, 7])))
    n3 = int(rs.choice([1, 3, 8, 8, 8, 9, 16, 32]))
    n4 = int(rs.choice([1, 4, 4, 4, 7, 32]))
    reg = REG_VARIANTS[rs.randint(4)]
    nd, ng = int(rs.randint(1, 13)), int(rs.randint(1, 13))
    torch.manual_seed(int(rs.randint(1 << 30)))
    net = RewardNet(d=d, reg=reg, k1=k1, f2=f2, k2=k2, n_fc3=n3, n_fc4=n4)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    net = net.to(dev)
    stores = []
    for cnt in (nd, ng):
        st = TrajectoryStore(d, 15, dev)
        extra = int(rs.randint(0, 4))
        st.push(torch.as_tensor(rs.dirichlet(np.ones(d), size=(extra + 1, 15)), dtype=torch.float32),
                torch.as_tensor(rs.dirichlet(np.ones(d), size=(extra + 1, 15, d)), dtype=torch.float32))
        st.push(torch.as_tensor(rs.dirichlet(np.ones(d) * rs.choice([0.3, 1.0, 4.0]), size=(cnt + 2, 15)), dtype=torch.float32),
                torch.as_tensor(rs.dirichlet(np.ones(d) * rs.choice([0.2, 1.0, 5.0]), size=(cnt + 2, 15, d)), dtype=torch.float32), drop=extra)
        stores.append(st)
    demo, gen = stores
    lr = 1e-3
    tr = RewardTrainer(net, lr)
    m = np.zeros(tr.flat.numel())
    v = np.zeros_like(m)
    fam = 'reference geometry (5, 3, 2)' if ref_geom else 'run-time geometry'
    w = worst.setdefault(fam, [0.0, 0.0, None, 0])
    for step in range(1, 4):
        di = list(rs.permutation(len(demo))[:nd])
        gi = list(rs.permutation(len(gen))[:ng])
        seed = int(rs.randint(1 << 62))
        prm = RO.params_from_torch(net)
        ds, da = batch_np(demo, di)
        gs, ga = batch_np(gen, gi)
        N = (nd + ng) * 15
        masks = RO.dropout_masks(net.keep_prob, seed, 0, N, n3, n4) if net.use_dropout else None
        (loss, first, second, regv), g, r = RO.irl_loss_and_grad(prm, ds, da, gs, ga, 5, ng, l1l2=net.use_l1l2, masks=masks)
        gflat = RO.flatten_like_kernel(g)
        # magnitude of the summed terms (|dL/dr| weights)
        state, action = np.concatenate([ds, gs], 0), np.concatenate([da, ga], 0)
        rr, cache = RO.forward_cache(prm, state, action, masks)
        S = rr[nd * 15:].reshape(ng, 15).sum(1)
        soft = np.exp(S - S.max()); soft /= soft.sum()
        dr = np.concatenate([np.full((nd * 15, 1), 0.2), np.repeat(soft, 15)[:, None]], 0)
        # (|weights| in the backward pass: every product of the chain rule enters with its magnitude -- the sum of |terms| an fp32
        #  result can be accurate against; the signed pass would hide the cancellation between pixels and units of one sample)
        scale = np.abs(RO.flatten_like_kernel(RO.backward({k_: np.abs(v_) for k_, v_ in prm.items()}, cache, dr)))
        if cache['kink'] < 1e-6:                 # a ReLU input within fp32 rounding (relative to its summed terms) of zero: the two evaluations may take different branches
            skipped += 1
            break
        p_dev = tr.flat.double().cpu().numpy()
        tr.step(demo, [demo.rows[i] for i in di], gen, [gen.rows[i] for i in gi], 5, seed, grad_only=True)
        got = tr.grad.double().cpu().numpy()
        offs = np.cumsum([0] + [q.numel() for q in net.parameters()])
        e_rel = 0.0
        for k in range(10):
            a_, b_, sc = got[offs[k]:offs[k + 1]], gflat[offs[k]:offs[k + 1]], scale[offs[k]:offs[k + 1]]
            e_rel = max(e_rel, float(np.max(np.abs(a_ - b_)) / max(np.max(np.abs(b_)), np.max(sc), 1e-3)))
        st = tr.stats.double().cpu().numpy()
        assert abs(st[0] - loss) <= 1e-5 * max(1.0, abs(loss)), (st[0], loss)
        # (d <= 21, the reference's sizes: 1e-5; the fp32 forward sums over 2 d^2 inputs grow with d: 5e-5 up to d = 32)
        if e_rel > (1e-5 if d <= 21 else 3e-5):
            per = [float(np.max(np.abs(got[offs[k]:offs[k + 1]] - gflat[offs[k]:offs[k + 1]])) / max(np.max(np.abs(gflat[offs[k]:offs[k + 1]])), np.max(scale[offs[k]:offs[k + 1]]), 1e-3)) for k in range(10)]
            rk = tr._ws[4:4 + N].double().cpu().numpy()
            print('FAIL', (e_rel, d, k1, f2, k2, n3, n4, reg, nd, ng), 'kink %.2e' % cache['kink'], ['%.0e' % e for e in per],
                  'max |r - r_oracle| %.2e' % float(np.max(np.abs(rk - r[:, 0]))), 'step', step)
            sys.exit(1)
        tr.step(demo, [demo.rows[i] for i in di], gen, [gen.rows[i] for i in gi], 5, seed)
        p_ref, m, v = RO.adam_tf(p_dev, gflat, m, v, step, lr=lr)
        gotp = tr.flat.double().cpu().numpy()
        # Adam's update is m / sqrt(v): sign-like at step 1, a ratio of gradient histories later -- a gradient entry known to
        # 1e-5 of the tensor's scale moves its update by (1e-5 scale / |g|) lr: compare where the entry is resolved to 1 %
        big = np.zeros(gflat.size, dtype=bool)
        for k in range(10):
            sl = slice(offs[k], offs[k + 1])
            big[sl] = np.abs(gflat[sl]) > 1e-3 * max(np.max(np.abs(gflat[sl])), np.max(scale[sl]))
        e_p = float(np.max(np.abs(gotp - p_ref)[big]) / lr) if big.any() else 0.0
        assert e_p <= 0.05, (e_p, d, n3, n4, reg)
        m, v = tr.m.double().cpu().numpy(), tr.v.double().cpu().numpy()
        if e_rel > w[0]:
            w[0], w[2] = e_rel, (d, k1, f2, k2, n3, n4, reg, nd, ng)
        w[1] = max(w[1], e_p)
    else:
        w[3] += 1
        n += 1
print('%d random configurations (3 updates each) in %.0f s: gradient within 1e-5 (d <= 21; 3e-5 up to d = 32) of the summed magnitude (sum of |terms| of the chain rule), loss within 1e-5, updated parameters within 0.05 lr of the fp64 oracle; %d draws skipped (a ReLU pre-activation within 1e-6, relative to its summed terms, of its kink)' % (n, time.time() - t0, skipped))
for fam in sorted(worst):
    print('  %-30s %4d configurations, largest gradient deviation %.2e at (d, k1, f2, k2, n3, n4, reg, n_demo, n_gen) = %s; largest parameter deviation %.3f lr'
          % (fam, worst[fam][3], worst[fam][0], worst[fam][2], worst[fam][1]))
