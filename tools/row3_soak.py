"""Randomised soak of the two lane mappings of the d = 21 (or, ROW3_SOAK_D=15, d = 15) sampling launches (developer tool; the claim of tests/test_gpu_row3.py
over many more shapes): the packed kernel k_core_small (three trajectories per wavefront, a lane per matrix row) and k_core_row3
(one trajectory per wavefront, three lanes per row) must write the SAME BITS -- pi trajectory, rewards, TD errors, scores, actions,
batch sums -- whatever the policy, the batch, the rollout length and the Philox keys are.
Random: batch 1 .. 4 500, T 1 .. 17, first step (odd / even), 64-bit trajectory offsets, theta 0.5 .. 40, shift, alpha_scale
0.3 .. 1e6 (shapes far below 1: the boost / exact paths; far above: the squeeze), start states from Dirichlet(0.05 .. 10) (one-hot-ish
to flat), gamma, TD on / off, WRITE_P, running discount, reward kind, start rows gathered / drawn in the kernel, a pending update.
Every output of mfg_rollout / mfg_train_rollout_deferred is compared with torch.equal.   usage: row3_soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops, _lib as L

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device('cuda:0')
lib = L.lib()
rs = np.random.RandomState(20261006)
d = int(os.environ.get("ROW3_SOAK_D", "21"))
F = ops.num_features(d)
t0 = time.time()
n = n_roll = n_def = 0
cold = 0
while time.time() - t0 < budget:
    B = int(rs.choice([rs.randint(1, 13), rs.randint(13, 300), rs.randint(300, 4500)]))
    T = int(rs.choice([1, 2, 3, 15, 15, rs.randint(1, 18)]))
    first_step = int(rs.randint(0, 1 << 20))
    traj_offset = int(rs.choice([0, rs.randint(1 << 30), (1 << 33) + rs.randint(1 << 30)]))
    theta = float(rs.choice([8.86349, rs.uniform(0.5, 40.0)]))
    shift = float(rs.choice([0.16, 0.0, rs.uniform(-0.3, 0.5)]))
    scale = float(rs.choice([12000.0, 1e4, 10 ** rs.uniform(-0.5, 6.0)]))
    conc = float(rs.choice([1.0, 0.05, 0.3, 10.0]))
    gamma = float(rs.choice([1.0, 0.9]))
    seed = int(rs.randint(1 << 31))
    th = torch.tensor([theta], dtype=torch.float64, device=dev)
    w = torch.as_tensor(rs.rand(F), device=dev)
    cfg = dict(B=B, T=T, first_step=first_step, traj_offset=traj_offset, theta=theta, shift=shift, scale=scale, conc=conc, seed=seed)
    if rs.rand() < 0.75:
        td = bool(rs.rand() < 0.8)
        write_P = bool(rs.rand() < 0.5)
        dpow = bool(td and rs.rand() < 0.3)
        kind = int(rs.choice([0, 0, 1])) if td else 0
        pi0 = torch.as_tensor(rs.dirichlet(conc * np.ones(d), size=B).astype(np.float32), device=dev)
        outs = []
        for mode in (1, 2):
            lib.mfg_set_core_mapping(mode)
            outs.append(ops.rollout(pi0, T, th, shift, scale, w=w if td else None, gamma=gamma, reward_kind=kind, seed=seed,
                                    first_step=first_step, traj_offset=traj_offset, td=td, write_P=write_P, discount_pow=dpow))
        keys = ['pi_traj', 'pi_last', 'reward', 'delta', 'g', 'P', 'G']
        cfg.update(td=td, write_P=write_P, dpow=dpow, kind=kind, call='rollout')
        n_roll += 1
    else:
        # a rank's multi-GPU cycle: start rows drawn in the kernel, a pending update applied in the weight staging
        mat = torch.as_tensor(rs.dirichlet(conc * np.ones(d), size=int(rs.randint(1, 70))).astype(np.float32), device=dev)
        outs = []
        for mode in (1, 2):
            lib.mfg_set_core_mapping(mode)
            G = torch.as_tensor(np.random.RandomState(seed).randn(F + 3) * 1e-3, device=dev)
            G[F + 2] = float(B * T)
            ta, wa = torch.empty_like(th), torch.empty_like(w)
            racc = torch.zeros(1, dtype=torch.float64, device=dev)
            bufs = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'pi_last': torch.empty(B, d, device=dev),
                    'reward': torch.empty(B, T, device=dev), 'delta': torch.empty(B, T, dtype=torch.float64, device=dev),
                    'g': torch.empty(B, T, dtype=torch.float64, device=dev)}
            ws = ops.workspace(B * T, d, dev)
            ops.train_rollout_deferred(mat, None, T, th, w, (G, 0.1, 0.001, racc.data_ptr()), ta, wa, shift, scale, gamma, G, ws, bufs,
                                       seed=seed, first_step=first_step, traj_offset=traj_offset)
            outs.append(dict(bufs, G=G.clone(), theta=ta, w=wa, racc=racc))
        keys = ['pi_traj', 'pi_last', 'reward', 'delta', 'g', 'G', 'theta', 'w', 'racc']
        cfg.update(call='train_rollout_deferred')
        n_def += 1
    a, b = outs
    for k in keys:
        x, y = a.get(k), b.get(k)
        if x is None and y is None:
            continue
        if not torch.equal(x, y):
            # (NaNs compare unequal: a policy far outside the mixed-precision range is reported by both kernels alike)
            if torch.isnan(x).any() and torch.equal(torch.isnan(x), torch.isnan(y)) and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y)):
                continue
            print('MISMATCH in %r: max |diff| %g   %s' % (k, float((x.double() - y.double()).abs().max()), cfg))
            lib.mfg_set_core_mapping(0)
            sys.exit(1)
    ops.clear_status()
    n += 1
lib.mfg_set_core_mapping(0)
print('%d random configurations in %.0f s (%d mfg_rollout, %d mfg_train_rollout_deferred): every output of k_core_row3 equals the packed '
      "kernel's bit for bit" % (n, time.time() - t0, n_roll, n_def))
print('OK')
