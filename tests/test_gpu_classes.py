"""-m gpu tests of the drop-in classes (mfg_ac2.actor_critic / ac_irl.AC_IRL call surface).

The seeded train() traces come from the unmodified reference (tests/golden/train_*.npz); with
rng='numpy', batch=1 the classes consume np.random in the reference's order, so theta after every step
must retrace the reference (fp32 storage of pi / P bounds the deviation)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU')
    return torch.device('cuda:0')


def AC(**kw):
    from discrete_mean_field_game_amd.mfg_ac2 import actor_critic
    kw.setdefault('verbose', 0)
    return actor_critic(**kw)


def IRL(**kw):
    from discrete_mean_field_game_amd.ac_irl import AC_IRL
    kw.setdefault('verbose', 0)
    return AC_IRL(**kw)


def O():
    from oracle import mfg_oracle
    return mfg_oracle


def fake_reward_dev(pi, P):
    """Device twin of oracle.gen_golden.fake_reward (closed-form stand-in of the TF net)."""
    diag = torch.diagonal(P, dim1=-2, dim2=-1).double()
    return torch.tanh(5.0 * (pi.double() * diag).sum(-1) - 0.3).float().contiguous()


# ---------------------------------------------------------------------------------------------------
def test_call_surface_shapes_and_values(dev):
    """The probes of the reference's test2.py, run against the drop-in class."""
    k = np.load(os.path.join(G, 'kat_mfg_ac2.npz'))
    np.random.seed(0)
    ac = AC(theta=10, shift=0.4, d=4, pi0=np.eye(4), rng='numpy')
    pi = np.array([0.7, 0.09, 0.01, 0.2])
    np.random.seed(42)
    P = ac.sample_action(pi)                                          # test2.py:14-32
    assert P.shape == (4, 4) and P.dtype == np.float64
    assert np.allclose(P, k['grad_P'], rtol=2e-6)                     # same legacy gamma stream, fp32 storage
    assert np.allclose(P.sum(1), 1, atol=1e-6)
    assert np.allclose(ac.mat_alpha, k['grad_alpha'], rtol=1e-6)
    assert np.allclose(ac.mat_alpha_deriv, k['grad_alpha_deriv'], rtol=1e-6, atol=1e-9)
    g = ac.calc_gradient_vectorized(P, pi)                            # test2.py:105-121
    assert isinstance(g, float) and abs(g - (-6.302201890992953)) < 1e-4
    # the reference's three-way self-check (test2.py:105-121): the two loop forms are evaluation paths of their own here
    # (mfg_alpha + torch digamma / log in fp64, no score kernel), the vectorised one is the HIP score kernel
    gb, gl = ac.calc_gradient_basic(P, pi), ac.calc_gradient(P, pi)
    assert isinstance(gb, float) and isinstance(gl, float)
    # (reference values of its own three variants on the fp64 P; the device holds P in fp32: 2e-6 relative on P)
    assert abs(gb - float(k['grad_basic'])) < 1e-4 and abs(gl - float(k['grad_loop'])) < 1e-4
    assert abs(gl - gb) < 1e-12 * abs(gb) + 1e-13
    assert abs(g - gb) < 1e-5 * abs(gb)                               # mixed-precision kernel vs the fp64 loop forms
    ac64 = AC(theta=10, shift=0.4, d=4, pi0=np.eye(4), rng='numpy', precision='f64')
    np.random.seed(42)
    P64 = ac64.sample_action(pi)
    g64 = ac64.calc_gradient_vectorized(P64, pi)
    assert abs(g64 - ac64.calc_gradient_basic(P64, pi)) < 1e-9 * abs(g64) and abs(g64 - ac64.calc_gradient(P64, pi)) < 1e-9 * abs(g64)
    r = ac.calc_reward(np.array([[1, 3, 3], [4, 5, 6], [7, 8, 9]]), np.array([0.1, 0.2, 0.7]), 3)   # test2.py:46-56
    assert r.shape == (1,) and abs(r[0] + 39.07) < 1e-4
    ac3 = AC(d=3, pi0=np.eye(3))
    ac3.w = np.ones(10)                                               # test2.py:73-88
    v = ac3.calc_value(np.array([0.1, 0.2, 0.7]))
    assert v.shape == (1,) and abs(v[0] - 2.77) < 1e-6
    f = ac3.calc_features(np.array([0.1, 0.2, 0.7]))
    assert np.allclose(f, [.01, .02, .07, .04, .14, .49, .1, .2, .7, 1.], rtol=1e-6)
    assert abs(ac3.JSD(np.array([.5, .5, 0.]), np.array([.1, .2, .7])) - 0.34858446189521375) < 1e-6
    # public mutable attributes
    ac3.theta = 7.5
    assert ac3.theta == 7.5 and ac3.w.shape == (10, 1) and ac3.num_start_samples == 3


def test_batched_inputs(dev):
    ac = AC(d=21, batch=8)
    rs = np.random.RandomState(0)
    pi = rs.dirichlet(np.ones(21), size=8)
    P = ac.sample_action(pi)
    assert P.shape == (8, 21, 21)
    assert ac.calc_reward(P, pi, 21).shape == (8,)
    assert ac.calc_value(pi).shape == (8,) and ac.calc_features(pi).shape == (8, 253)
    assert ac.calc_gradient_vectorized(P, pi).shape == (8,)
    Pt = ac.sample_action(torch.as_tensor(pi, device=dev, dtype=torch.float32))
    assert isinstance(Pt, torch.Tensor) and Pt.shape == (8, 21, 21)


@pytest.mark.parametrize('name', ['c0_g1', 'c1_g1', 'c0_g09', 'c1_g09'])
@pytest.mark.parametrize('precision', ['mixed', 'f64'])
def test_train_retraces_reference_mfg_ac2(dev, name, precision):
    z = np.load(os.path.join(G, 'train_mfg_ac2_%s.npz' % name))
    np.random.seed(int(z['seed']))
    ac = AC(theta=float(z['theta0']), shift=float(z['shift']), alpha_scale=float(z['alpha_scale']), d=21,
            pi0=z['mat_pi0'], batch=1, rng='numpy', update_every='step', precision=precision)
    assert np.array_equal(ac.w, z['w0'])                              # same init_w draw as the reference
    ac.trace = []
    ac.train(num_episodes=int(z['num_episodes']), gamma=float(z['gamma']), constant=int(z['constant']),
             lr_critic=float(z['lr_critic']), lr_actor=float(z['lr_actor']), consecutive=100)
    ref = np.concatenate([z['theta_before'][1:], [float(z['theta_final'])]])
    got = np.array(ac.trace)
    assert got.shape == ref.shape
    # theta after every one of the 90 updates; the increments themselves to 1e-5 relative
    assert np.max(np.abs(got - ref)) < 2e-7
    inc_ref = np.diff(np.concatenate([[float(z['theta0'])], ref]))
    inc_got = np.diff(np.concatenate([[float(z['theta0'])], got]))
    big = np.abs(inc_ref) > 1e-6
    assert np.max(np.abs(inc_got[big] - inc_ref[big]) / np.abs(inc_ref[big])) < 2e-4
    assert np.max(np.abs(ac.w - z['w_final'])) < 1e-6
    assert isinstance(ac.theta, np.ndarray) and ac.theta.shape == (1,)


@pytest.mark.parametrize('name', ['c0_g1', 'c1_g09', 'c0_g09_stop'])
def test_train_retraces_reference_ac_irl(dev, name):
    z = np.load(os.path.join(G, 'train_ac_irl_%s.npz' % name))
    np.random.seed(int(z['seed']))
    ac = IRL(theta=float(z['theta0']), shift=float(z['shift']), alpha_scale=float(z['alpha_scale']), d=21,
             pi0=z['mat_pi0'], batch=1, rng='numpy', num_policies=3, use_tf=False)
    assert np.array_equal(ac.w, z['w0'])
    ac.trace = []
    ac.train(max_episodes=int(z['max_episodes']), stop_criteria=float(z['stop_criteria']), gamma=float(z['gamma']),
             constant=bool(z['constant']), lr_critic=float(z['lr_critic']), lr_actor=float(z['lr_actor']),
             consecutive=100, reward_fn=fake_reward_dev)
    n = int(z['steps_run'])
    assert len(ac.trace) == n and ac.episodes_run == n // 15          # same early-stop episode
    ref = np.concatenate([z['theta_before'][1:], [float(z['theta_final'])]])
    assert np.max(np.abs(np.array(ac.trace) - ref)) < 2e-6
    assert np.max(np.abs(ac.w - z['w_final'])) < 1e-5
    lp = np.array([float(np.ravel(t)[0]) for t in ac.list_policies])
    assert np.allclose(lp, z['list_policies'], atol=2e-6)             # policy FIFO (ac_irl.py:731)


def test_philox_step_mode_matches_oracle_replay(dev):
    """update_every='step', B>1: every update equals the oracle's batch-mean update on the sampled actions."""
    from discrete_mean_field_game_amd import ops
    d, B = 21, 6
    rs = np.random.RandomState(3)
    mat = rs.dirichlet(np.ones(d), size=5)
    np.random.seed(11)
    ac = AC(d=d, pi0=mat, batch=B, rng='philox', seed=77, update_every='step', precision='f64')
    w0 = ac.w[:, 0].copy(); theta0 = float(ac.theta)
    ac.trace = []
    np.random.seed(12)
    ac.train(num_episodes=1, gamma=0.9, constant=0)
    # replay: same start draw (Philox, keyed by seed / first step of the episode / trajectory id), same action counters,
    # oracle math in fp64
    from oracle.philox_ref import start_indices
    idx = start_indices(77, 0, np.arange(B), 5)
    pi = mat[idx].astype(np.float32)
    w = w0.copy(); theta = theta0
    F = O().num_features(d)
    for t in range(15):
        th = torch.tensor([theta], dtype=torch.float64, device=dev)
        P = ops.sample_dirichlet(torch.as_tensor(pi, device=dev), th, 0.16, 12000.0, seed=77, step=t,
                                 precision='f64').cpu().numpy()
        pn = O().transition(P, pi).astype(np.float32)
        r = O().calc_reward(P.astype(np.float64), pi.astype(np.float64))
        delta, g, G_w, G_theta, _ = O().batched_td_pg(pi, pn, P, r, w, theta, 0.16, 0.9)
        sc, sa = O().lr_scales(0, False)
        w = w + 0.1 * sc * G_w / B
        theta = theta + 0.001 * sa * G_theta / B
        assert abs(ac.trace[t] - theta) < 1e-9, t
        pi = pn
    assert np.max(np.abs(ac.w[:, 0] - w)) < 1e-9


def test_rollout_mode_and_determinism(dev):
    d, B = 21, 512
    mat = np.random.RandomState(0).dirichlet(np.ones(d), size=16)
    outs = []
    for _ in range(2):
        np.random.seed(5)
        ac = AC(d=d, pi0=mat, batch=B, seed=9, update_every='rollout')
        ac.train(num_episodes=4, consecutive=2)
        outs.append((float(ac.theta[0]), ac.w.copy()))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])     # bitwise reproducible
    assert np.isfinite(outs[0][0]) and outs[0][0] != 8.86349


def test_generate_trajectory_and_evaluate(dev, tmp_path, monkeypatch):
    d = 21
    rs = np.random.RandomState(2)
    ac = AC(d=d, pi0=rs.dirichlet(np.ones(d), size=4), seed=3)
    tr = ac.generate_trajectory(ac.mat_pi0[1], 16)
    assert tr.shape == (16, d) and np.allclose(tr[0], ac.mat_pi0[1].astype(np.float32), atol=0)
    assert np.allclose(tr.sum(1), tr[0].sum(), atol=5e-6)
    trb = ac.generate_trajectory(ac.mat_pi0, 16)
    assert trb.shape == (4, 16, d)
    # evaluate() over files in the reference's on-disk format (mfg_ac2.py:137)
    monkeypatch.chdir(tmp_path)
    os.makedirs('test_normalized_round2'); os.makedirs('eval_mfg_round2')
    for day in range(22, 25):
        np.savetxt('test_normalized_round2/trend_distribution_day%d.csv' % day, rs.dirichlet(np.ones(d + 2), size=16),
                   fmt='%.3e', delimiter=' ')
    res = ac.evaluate(theta=8.86349, shift=0.5, alpha_scale=1e4, d=d, outfile='eval_mfg_round2/out.csv', write_header=1)
    assert len(res) == 4 and all(np.isfinite(res)) and res[2] >= 0
    lines = open('eval_mfg_round2/out.csv').read().strip().split('\n')
    assert lines[0].startswith('theta,shift,alpha_scale') and len(lines[1].split(',')) == 11
    best = ac.gridsearch([8.0, 9.0], [0.5], [1e4], indir='test_normalized_round2', outfile='eval_mfg_round2/out.csv')
    assert len(best) == 4


def test_numpy_rng_generate_trajectory_retraces_reference(dev):
    z = np.load(os.path.join(G, 'generate_trajectory_mfg_ac2.npz'))
    np.random.seed(0)
    ac = AC(theta=float(z['theta']), shift=float(z['shift']), alpha_scale=float(z['alpha_scale']), d=21,
            pi0=z['pi0'][None], rng='numpy')
    np.random.seed(int(z['seed']))
    tr = ac.generate_trajectory(z['pi0'], int(z['total_hours']))
    assert np.allclose(tr, z['traj'], rtol=5e-6, atol=1e-9)


def test_irl_outerloop_smoke(dev):
    d = 15
    rs = np.random.RandomState(4)
    mat = rs.dirichlet(np.ones(d), size=6)
    demos = [[(rs.dirichlet(np.ones(d)), rs.dirichlet(np.ones(d), size=d)) for _ in range(15)] for _ in range(7)]
    np.random.seed(1)
    torch.manual_seed(1)
    ac = IRL(d=d, pi0=mat, demonstrations=demos, batch=64, num_policies=2, seed=5)
    tr = ac.generate_trajectories(3)
    assert len(tr) == 3 and len(tr[0]) == 15 and tr[0][0][1].shape == (d, d)
    assert np.allclose(tr[0][1][0], tr[0][0][1].T.dot(tr[0][0][0]), rtol=1e-5, atol=1e-7)   # pi' = P^T pi
    r = ac.reward(torch.rand(8, d, device=dev), torch.rand(8, d, d, device=dev))
    assert r.shape == (8,) and float(r.abs().max()) < 1
    ac.list_generated = ac.generate_trajectories(10)
    ac.list_eval_gen_transitions = [p for t in ac.list_generated for p in t]
    before = [p.detach().clone() for p in ac.reward_net.parameters()]
    ac.reward_iteration(max_iterations=10, stop_criteria=-1, iter_check=5)
    assert any(not torch.equal(a, b) for a, b in zip(before, ac.reward_net.parameters()))
    assert np.isfinite(ac.loss_val)
    ac.train(max_episodes=3, stop_criteria=-1)
    assert len(ac.list_policies) == 2 and np.isfinite(float(np.ravel(ac.theta)[0]))


@pytest.mark.parametrize('reg', ['none', 'dropout_l1l2'])
@pytest.mark.parametrize('d,n3,n4', [(15, 8, 4), (21, 8, 4), (21, 6, 8), (32, 4, 4), (4, 3, 2), (21, 16, 4), (15, 16, 8), (21, 20, 4),
                                     (15, 17, 3)])
def test_reward_net_hip_forward_matches_torch_and_numpy(dev, reg, d, n3, n4):
    """One-launch HIP forward of the reward net (ac_irl.py:683 batched) vs the PyTorch module and the NumPy
    restatement of networks.py:46-81 (dropout off: deterministic part)."""
    from discrete_mean_field_game_amd import ops
    from discrete_mean_field_game_amd.networks import RewardNet
    from oracle import reward_net_oracle as RO
    torch.manual_seed(d + n3)
    net = RewardNet(d=d, reg=reg, n_fc3=n3, n_fc4=n4, dropout_always=False).to(dev).eval()
    for p in net.parameters():
        if p.dim() == 1:
            torch.nn.init.uniform_(p, -0.2, 0.2)
    assert ops.reward_net_supported(net)
    rs = np.random.RandomState(d)
    B = 333
    state = rs.dirichlet(np.ones(d), size=B).astype(np.float32)
    action = rs.dirichlet(np.ones(d), size=(B, d)).astype(np.float32)
    s_t, a_t = torch.as_tensor(state, device=dev), torch.as_tensor(action, device=dev)
    out = ops.reward_net_forward(net, s_t, a_t).cpu().numpy()
    with torch.no_grad():
        ref_t = net(s_t, a_t).reshape(-1).cpu().numpy()
    ref_n = RO.forward(RO.params_from_torch(net), state.astype(np.float64), action.astype(np.float64))[:, 0]
    assert out.shape == (B,)
    assert np.max(np.abs(out - ref_n)) < 2e-6                  # fp32 kernel vs fp64 restatement
    assert np.max(np.abs(out - ref_t)) < 5e-6                  # vs the MIOpen / rocBLAS path
    assert np.max(np.abs(out)) < 1


@pytest.mark.parametrize('d,n3,n4', [(21, 8, 4), (15, 8, 4), (32, 32, 32), (21, 6, 8), (21, 16, 16), (21, 24, 4)])
def test_reward_net_hip_dropout_masks_are_the_documented_philox_bits(dev, d, n3, n4):
    """With dropout ON (the reference's default reg and its behaviour when the net serves as the RL reward) the kernel's
    output equals the oracle evaluated with masks redrawn from the documented counters: unit o of FC3 / FC4 of sample n
    <- Philox4x32-10(key = seed; counter = (o, 3 | 4, sample_offset + n, 0)).  Any (seed, sample_offset), incl. offsets
    beyond 2^32 (the high trajectory bits of the counter)."""
    from discrete_mean_field_game_amd import ops
    from discrete_mean_field_game_amd.networks import RewardNet
    from oracle import reward_net_oracle as RO
    torch.manual_seed(d + n4)
    net = RewardNet(d=d, reg='dropout_l1l2', n_fc3=n3, n_fc4=n4).to(dev).eval()
    for p in net.parameters():
        if p.dim() == 1:
            torch.nn.init.uniform_(p, -0.2, 0.2)
    rs = np.random.RandomState(d)
    B = 257
    state = rs.dirichlet(np.ones(d), size=B).astype(np.float32)
    action = rs.dirichlet(np.ones(d), size=(B, d)).astype(np.float32)
    s_t, a_t = torch.as_tensor(state, device=dev), torch.as_tensor(action, device=dev)
    params = RO.params_from_torch(net)
    for seed, off in ((77, 0), (0x9E3779B97F4A7C15, 1000), (5, (1 << 33) + 12345)):
        out = ops.reward_net_forward(net, s_t, a_t, seed=seed, sample_offset=off).cpu().numpy()
        ref = RO.forward(params, state.astype(np.float64), action.astype(np.float64), dropout=(0.4, seed, off))[:, 0]
        assert np.max(np.abs(out - ref)) < 5e-6, (seed, off)
        m3, m4 = RO.dropout_masks(0.4, seed, off, B, n3, n4)
        assert 0.3 < (m3 > 0).mean() < 0.5 and 0.25 < (m4 > 0).mean() < 0.55       # keep probability 0.4


@pytest.mark.parametrize('d', [21, 15])
@pytest.mark.parametrize('B', [1, 15, 16, 17, 4111, 8192 + 5])
def test_reward_net_hip_group_edges(dev, d, B):
    """The matrix-core kernel evaluates 16 samples per block pass (one per wave, FC3 split along K over the 16 waves):
    batches that are not multiples of 16, fewer samples than one group, more groups than resident blocks (256) -- with
    and without dropout, every sample against the oracle."""
    from discrete_mean_field_game_amd import ops
    from discrete_mean_field_game_amd.networks import RewardNet
    from oracle import reward_net_oracle as RO
    torch.manual_seed(B)
    net = RewardNet(d=d, reg='dropout_l1l2').to(dev).eval()
    rs = np.random.RandomState(B + d)
    state = rs.dirichlet(np.ones(d), size=B).astype(np.float32)
    action = rs.dirichlet(np.ones(d), size=(B, d)).astype(np.float32)
    s_t, a_t = torch.as_tensor(state, device=dev), torch.as_tensor(action, device=dev)
    params = RO.params_from_torch(net)
    out = ops.reward_net_forward(net, s_t, a_t, dropout=False).cpu().numpy()
    ref = RO.forward(params, state.astype(np.float64), action.astype(np.float64))[:, 0]
    assert out.shape == (B,) and np.max(np.abs(out - ref)) < 2e-6
    out = ops.reward_net_forward(net, s_t, a_t, seed=9, sample_offset=77).cpu().numpy()
    ref = RO.forward(params, state.astype(np.float64), action.astype(np.float64), dropout=(0.4, 9, 77))[:, 0]
    assert np.max(np.abs(out - ref)) < 5e-6


def test_reward_net_hip_weights_at_odd_addresses(dev):
    """The matrix-core kernel fetches the FC3 weights in 16-byte pieces from rows that start at multiples of 8 bytes; a
    weight buffer that is only 4-byte aligned takes the other kernels -- same results."""
    from discrete_mean_field_game_amd import ops
    from discrete_mean_field_game_amd.networks import RewardNet
    from oracle import reward_net_oracle as RO
    torch.manual_seed(3)
    d, B = 21, 100
    net = RewardNet(d=d, reg='none').to(dev).eval()
    rs = np.random.RandomState(2)
    state = rs.dirichlet(np.ones(d), size=B).astype(np.float32)
    action = rs.dirichlet(np.ones(d), size=(B, d)).astype(np.float32)
    s_t, a_t = torch.as_tensor(state, device=dev), torch.as_tensor(action, device=dev)
    ref = RO.forward(RO.params_from_torch(net), state.astype(np.float64), action.astype(np.float64))[:, 0]
    out0 = ops.reward_net_forward(net, s_t, a_t).cpu().numpy()
    w = net.fc3.weight.data
    buf = torch.empty(w.numel() + 1, dtype=w.dtype, device=dev)
    buf[1:].copy_(w.reshape(-1))
    net.fc3.weight.data = buf[1:].view_as(w)                         # contiguous, 4 bytes past a 16-byte boundary
    assert net.fc3.weight.data_ptr() % 8 == 4 and net.fc3.weight.is_contiguous()
    out1 = ops.reward_net_forward(net, s_t, a_t).cpu().numpy()
    assert np.max(np.abs(out0 - ref)) < 2e-6 and np.max(np.abs(out1 - ref)) < 2e-6


def test_reward_net_hip_dropout_statistics(dev):
    """Dropout (keep 0.4, inverted scaling) stays on when the net is the RL reward, like the reference
    (tf.contrib.layers.dropout defaults to is_training=True): masks differ per call and per sample, and the
    pre-activation mean is preserved."""
    from discrete_mean_field_game_amd import ops
    from discrete_mean_field_game_amd.networks import RewardNet
    torch.manual_seed(0)
    d = 21
    net = RewardNet(d=d, reg='dropout_l1l2').to(dev)
    with torch.no_grad():
        net.out.weight.mul_(0.05)                                # keep tanh in its linear range
    rs = np.random.RandomState(1)
    s1 = rs.dirichlet(np.ones(d)).astype(np.float32); a1 = rs.dirichlet(np.ones(d), size=d).astype(np.float32)
    B = 20000
    s_t = torch.as_tensor(np.repeat(s1[None], B, 0), device=dev)
    a_t = torch.as_tensor(np.repeat(a1[None], B, 0), device=dev)
    r1 = ops.reward_net_forward(net, s_t, a_t, seed=1).cpu().numpy()
    r2 = ops.reward_net_forward(net, s_t, a_t, seed=2).cpu().numpy()
    r1b = ops.reward_net_forward(net, s_t, a_t, seed=1).cpu().numpy()
    assert np.array_equal(r1, r1b) and not np.array_equal(r1, r2)       # counter-based masks
    assert len(np.unique(np.round(r1, 6))) > 10                          # masks differ across samples
    with torch.no_grad():
        torch.manual_seed(3)
        rt = net(s_t, a_t).reshape(-1).cpu().numpy()                     # torch's dropout, same distribution
    assert abs(r1.mean() - rt.mean()) < 4 * (r1.std() + rt.std()) / np.sqrt(B) + 1e-4
    assert abs(r1.std() - rt.std()) < 0.05 * rt.std() + 1e-5
    r0 = ops.reward_net_forward(net, s_t[:4].contiguous(), a_t[:4].contiguous(), dropout=False).cpu().numpy()
    assert np.allclose(r0, r0[0])


def test_backward_value_kernel_vs_reference_golden(dev):
    """mfg_synthetic backward recursion V^n = r + P V^{n+1} and the evaluate_synthetic(_JSD) metrics on the
    actions captured from the unmodified reference (fixture pinned to its returned mean/std)."""
    from discrete_mean_field_game_amd import ops
    z = np.load(os.path.join(G, 'backward_value_mfg_synthetic.npz'))
    for key, mean_k, std_k, pick in (('actions_l1', 'l1_mean', 'l1_std', 1), ('actions_jsd', 'jsd_mean', 'jsd_std', 2)):
        acts = z[key].astype(np.float32)
        out = ops.backward_value(torch.as_tensor(acts, device=dev))
        Vo, l1o, jso = O().evaluate_synthetic_diffs(acts)                 # oracle on identical fp32 inputs
        assert np.allclose(out[0].cpu().numpy(), Vo, rtol=1e-12, atol=1e-13)
        assert np.allclose(out[1].cpu().numpy(), l1o, rtol=1e-12)
        assert np.allclose(out[2].cpu().numpy(), jso, rtol=1e-10)
        vals = out[pick].cpu().numpy()
        assert abs(vals.mean() - float(z[mean_k])) < 1e-5 * abs(float(z[mean_k]))     # vs the reference (fp64 actions)
        assert abs(vals.std() - float(z[std_k])) < 1e-4 * abs(float(z[std_k])) + 1e-9


def test_mfg_synthetic_class_surface(dev):
    from discrete_mean_field_game_amd.mfg_synthetic import actor_critic as SAC
    z = np.load(os.path.join(G, 'reward_mfg_synthetic.npz'))
    np.random.seed(0)
    ac = SAC(theta=2.6, shift=0.0, alpha_scale=10000, d=21, pi0=z['pi'], verbose=0)
    r = ac.calc_reward(z['P'], z['pi'], 21)
    assert np.allclose(r, z['reward'], rtol=1e-5)
    kat = ac.calc_reward(np.array([[1, 3, 3], [4, 5, 6], [7, 8, 9.]]), np.array([.1, .2, .7]), 3)
    assert abs(kat[0] - float(z['kat_reward'])) < 1e-4 and abs(kat[0] + 76.55) < 1e-4
    traj, acts = ac.generate_trajectory(z['pi'][0], 16)
    assert traj.shape == (16, 21) and acts.shape == (15, 21, 21)
    assert np.allclose(acts[0].T.dot(traj[0]), traj[1], rtol=1e-5, atol=1e-7)
    assert np.allclose(ac.calc_reward_vector(acts[0]), O().calc_reward_vector(acts[0]), rtol=1e-6)
    m, s = ac.evaluate_synthetic(day_first=1, day_last=4)
    mj, sj = ac.evaluate_synthetic_JSD(day_first=1, day_last=4)
    assert np.isfinite([m, s, mj, sj]).all() and m > 0 and ac.mat_V.shape == (4, 16, 21)
    assert abs(ac.JSD(np.array([.5, .5, -1.]), np.array([.1, .2, .7])) - O().JSD_synthetic([.5, .5, -1.], [.1, .2, .7])) < 1e-6
    ac.train(num_episodes=2, constant=1)                                  # synthetic reward inside the fused path
    assert np.isfinite(float(np.ravel(ac.theta)[0]))


def test_mfg_synthetic_train_logs_w(dev, tmp_path):
    """mfg_synthetic.train (mfg_synthetic.py:426-522): `results_syn/` defaults, the extra `file_w` keyword and one w line
    ('%.5e', comma separated, like train_log :419-423) per report next to theta / pi / reward."""
    import inspect
    from discrete_mean_field_game_amd.mfg_synthetic import actor_critic as SAC
    sig = inspect.signature(SAC.train).parameters
    assert sig['file_w'].default == 'results_syn/w.csv' and sig['file_theta'].default == 'results_syn/theta.csv'
    z = np.load(os.path.join(G, 'reward_mfg_synthetic.npz'))
    np.random.seed(1)
    ac = SAC(theta=2.6, shift=0.0, alpha_scale=10000, d=21, pi0=z['pi'], verbose=0)
    files = {k: str(tmp_path / (k + '.csv')) for k in ('theta', 'pi', 'reward', 'w')}
    ac.train(num_episodes=4, constant=1, consecutive=2, write_file=1, file_theta=files['theta'], file_pi=files['pi'],
             file_reward=files['reward'], file_w=files['w'])
    lines = {k: open(v).read().splitlines() for k, v in files.items()}
    assert [len(v) for v in lines.values()] == [2, 2, 2, 2]              # reports at episodes 0 and 2 (0-indexed, :513)
    last = lines['w'][-1].split(',')
    assert len(last) == 21 * 22 // 2 + 21 + 1
    assert all(len(x.split('e')[0]) == len('%.5e' % 1.0) - 4 or x.startswith('-') for x in last)
    # the run made two more updates after the last report; the logged w is a genuinely earlier state of the same vector
    assert not np.allclose(np.array(last, dtype=np.float64), np.ravel(ac.w), rtol=0, atol=0)
    assert np.allclose(np.array(last, dtype=np.float64), np.ravel(ac.w), rtol=0.5, atol=0.5)


def test_irl_importance_weights_calc_z(dev):
    """AC_IRL.calc_z / calc_pdf_action (ac_irl.py:270-379) against the oracle's log-space restatement."""
    from discrete_mean_field_game_amd import ac_irl
    from oracle import mfg_oracle as O
    rs = np.random.RandomState(3)
    d = 15
    ac = ac_irl.AC_IRL(d=d, pi0=rs.dirichlet(np.ones(d), size=8), demonstrations=[], seed=5, verbose=0)
    ac.list_policies = [8.64, 7.9, 8.2]
    trajs = ac.generate_trajectories(4)
    lz = ac.calc_z(trajs, log=True)
    pis = np.array([[p[0] for p in t] for t in trajs]); Ps = np.array([[p[1] for p in t] for t in trajs])
    want = O.calc_z(pis.astype(np.float32), Ps.astype(np.float32), ac.list_policies, ac.shift, ac.num_start_samples)
    np.testing.assert_allclose(lz, want, rtol=1e-9, atol=1e-6)
    z = ac.calc_z(trajs)
    assert z.shape == (4,) and np.all(z >= 0)
    lq = ac.calc_pdf_action(8.64, trajs[0][0][1], trajs[0][0][0], log=True)
    assert abs(lq - O.policy_logpdf(np.float32(trajs[0][0][0])[None], np.float32(trajs[0][0][1])[None], [8.64], ac.shift)[0, 0]) < 1e-6 * abs(lq)


@pytest.mark.parametrize('d,B,precision', [(21, 300, 'mixed'), (21, 7, 'f64'), (15, 64, 'mixed'), (47, 20, 'mixed'), (100, 5, 'mixed')])
def test_native_episode_loop_equals_python_step_loop(dev, d, B, precision):
    """mfg_train_episode (the whole 15-step episode with per-step updates issued natively) gives bit for bit what
    the per-step Python sequence rollout(T=1) -> apply_update gives (same kernels, same order, same Philox steps)."""
    rs = np.random.RandomState(d)
    mat = rs.dirichlet(np.ones(d), size=9)
    runs = []
    for use_native in (True, False):
        np.random.seed(21)
        ac = AC(d=d, pi0=mat, batch=B, rng='philox', seed=5, update_every='step', precision=precision, verbose=0)
        if not use_native:
            ac.trace = []                       # tracing forces the per-step Python path
        np.random.seed(22)
        ac.train(num_episodes=3, gamma=0.95, constant=0, consecutive=2)
        runs.append((np.ravel(ac.theta).copy(), ac.w.copy(), ac._last_pi.cpu().numpy().copy(), ac._rng_step))
    assert runs[1][3] == runs[0][3] == 45
    assert np.array_equal(runs[0][0], runs[1][0])
    assert np.array_equal(runs[0][1], runs[1][1])
    assert np.array_equal(runs[0][2], runs[1][2])


@pytest.mark.parametrize('d,B,precision,reg', [(21, 300, 'mixed', 'dropout_l1l2'), (21, 4096, 'mixed', 'dropout_l1l2'),
                                               (15, 64, 'f64', 'l1l2'), (21, 7, 'f64', 'dropout')])
def test_native_irl_episode_equals_python_step_loop(dev, d, B, precision, reg):
    """mfg_train_episode_irl (ac_irl.py:664-712: per env step sample + transition + score | reward network | batch sums +
    update, the whole 15-step episode issued natively) gives what the per-step Python sequence rollout(T=1, EXTERNAL)
    -> reward() -> grad_apply gives: same sampling / reward kernels, same order, same Philox steps and the same
    dropout-mask keys (dropout stays ON in the reference when the net serves as the RL reward).  The one difference is
    the association of the fp64 batch sums: the native loop forms them inside the reward-network launch (a row per block
    of eight samples), the Python sequence in the gradient kernel -- so the parameters agree to ~1e-13, not bit for bit."""
    import random
    rs = np.random.RandomState(d + B)
    mat = rs.dirichlet(np.ones(d), size=9)
    runs = []
    for use_native in (True, False):
        np.random.seed(21); torch.manual_seed(21); random.seed(21)
        ac = IRL(d=d, pi0=mat, demonstrations=[], batch=B, rng='philox', seed=5, update_every='step', precision=precision,
                 reg=reg, verbose=0)
        with torch.no_grad():
            for p in ac.reward_net.parameters():
                if p.dim() == 1:
                    p.uniform_(-0.2, 0.2)
        if not use_native:
            ac.trace = []                       # tracing forces the per-step Python path
        np.random.seed(22)
        ac.train(max_episodes=3, stop_criteria=-1, gamma=0.95, constant=False, consecutive=2)
        runs.append((np.ravel(ac.theta).copy(), ac.w.copy(), ac._rng_step, ac._reward_calls))
    assert runs[0][2] == runs[1][2] == 45 and runs[0][3] == runs[1][3] == 45
    assert abs(runs[0][0][0] - runs[1][0][0]) <= 1e-12 * abs(runs[1][0][0])
    assert np.max(np.abs(runs[0][1] - runs[1][1])) <= 1e-12 * np.max(np.abs(runs[1][1]))
    assert runs[0][0][0] != 8.64


@pytest.mark.parametrize('d,B,T,n3,precision', [(21, 300, 15, 8, 'mixed'), (21, 4096, 2, 8, 'mixed'), (21, 50, 1, 8, 'f64'),
                                                (15, 130, 4, 16, 'mixed'), (21, 90, 3, 24, 'mixed'), (12, 40, 3, 8, 'mixed')])
def test_irl_episode_with_drawn_start_states_equals_draw_then_episode(dev, d, B, T, n3, precision):
    """mfg_train_episode_irl_draw (start states drawn inside the first step kernel of the two-launch flow; by a draw launch in the
    three-launch flow: n_fc3 = 24 takes the run-mapped network kernel, d = 12 the generic one) == mfg_draw_start followed by
    mfg_train_episode_irl, bit for bit: parameters, batch sums, return, final states, the last step's P / reward / delta / g.
    Odd and even T (the ping-pong buffers end in pi_out without a copy), T = 1 (no step kernel with rows at all)."""
    from discrete_mean_field_game_amd import ops
    from discrete_mean_field_game_amd.networks import RewardNet
    ops.init()
    rs = np.random.RandomState(d * 1000 + B + T)
    torch.manual_seed(d + B)
    mat = torch.as_tensor(rs.dirichlet(np.ones(d), size=11).astype(np.float32), device=dev)
    net = RewardNet(d=d, reg='dropout_l1l2', n_fc3=n3, n_fc4=8).to(dev)
    F = ops.num_features(d)
    w0 = rs.rand(F) * 0.1
    outs = []
    for draw_inside in (False, True):
        th = torch.tensor([8.64], dtype=torch.float64, device=dev)
        w = torch.as_tensor(w0.copy(), device=dev)
        G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
        racc = torch.zeros(1, dtype=torch.float64, device=dev)
        ws = ops.workspace(B, d, dev)
        bufs = dict(ops.episode_buffers(B, d, dev), P=torch.empty(B, d, d, dtype=torch.float32, device=dev))
        if draw_inside:
            pi = torch.full((B, d), float('nan'), dtype=torch.float32, device=dev)
            kw = dict(mat_pi0=mat)
        else:
            _, pi = ops.draw_start(mat, B, 7, 40, 5)
            kw = {}
        ops.train_episode_irl(pi, T, th, 0.1, 1e4, w, 0.95, 0.1, 0.001, net, G, ws, bufs, seed=7, first_step=40, traj_offset=5,
                              rn_seed=99, rn_call0=3, rn_sample_offset=5, reward_acc=racc, precision=precision, **kw)
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy().copy() for t in (th, w, G, racc, pi, bufs['P'], bufs['reward'], bufs['delta'], bufs['g'])])
    for x, y in zip(*outs):
        assert np.array_equal(x, y)
    assert np.isfinite(outs[1][4]).all() and abs(outs[1][4].sum(axis=1) - 1).max() < 1e-5
    assert outs[1][0][0] != 8.64 and outs[1][2][F + 2] == B


def test_training_step_is_hip_graph_capturable(dev):
    """The C ABI launches on the caller's stream and never allocates or synchronises (after mfg_init), so a whole
    update (fused TD rollout + gradient kernels + parameter update) can be captured into a HIP graph and replayed;
    replays are bit-identical to eager execution."""
    from discrete_mean_field_game_amd import ops
    ops.init()
    d, B, T = 21, 512, 15
    rs = np.random.RandomState(0)
    pi0 = torch.as_tensor(rs.dirichlet(np.ones(d), size=B).astype(np.float32), device=dev)
    F = ops.num_features(d)
    w0 = rs.rand(F)

    def make():
        th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
        w = torch.as_tensor(w0.copy(), device=dev)
        G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
        ws = ops.workspace(B * T, d, dev)
        bufs = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'pi_last': torch.empty(B, d, device=dev),
                'reward': torch.empty(B, T, device=dev), 'delta': torch.empty(B, T, dtype=torch.float64, device=dev),
                'g': torch.empty(B, T, dtype=torch.float64, device=dev)}

        def step():
            ops.rollout(pi0, T, th, 0.16, 12000.0, w=w, gamma=1.0, seed=3, td=True, G=G, ws=ws, out=bufs)
            ops.apply_update(G, d, 0.1, 0.001, w, th)
        return th, w, step

    th_e, w_e, eager = make()
    for _ in range(3):
        eager()
    th_g, w_g, gstep = make()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            gstep()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(th_e, th_g) and torch.equal(w_e, w_g)
    assert float(th_g) != 8.86349


def test_gridsearch_equals_pointwise_evaluate(dev, tmp_path, monkeypatch):
    """gridsearch (mfg_ac2.py:673-689) evaluates every grid point on the device without per-point host round trips;
    its CSV lines and its argmin table are those of calling evaluate() point by point."""
    d = 21
    rs = np.random.RandomState(4)
    monkeypatch.chdir(tmp_path)
    os.makedirs('test_normalized_round2'); os.makedirs('out')
    for day in range(22, 27):
        np.savetxt('test_normalized_round2/trend_distribution_day%d.csv' % day, rs.dirichlet(np.ones(d), size=16),
                   fmt='%.3e', delimiter=' ')
    thetas, shifts, alphas = [6.0, 8.0, 9.5], [0.1, 0.5], [1e3, 1e4]
    a = AC(d=d, pi0=rs.dirichlet(np.ones(d), size=4), seed=9, verbose=0)
    best = a.gridsearch(thetas, shifts, alphas, indir='test_normalized_round2', outfile='out/grid.csv')
    b = AC(d=d, pi0=a.mat_pi0, seed=9, verbose=0)
    ref = [[100, 0, 0, 0] for _ in range(4)]
    for th in thetas:
        for sh in shifts:
            for al in alphas:
                r = b.evaluate(th, sh, al, d=d, indir='test_normalized_round2', outfile='out/point.csv')
                for k in range(4):
                    if r[k] <= ref[k][0]:
                        ref[k] = [r[k], th, sh, al]
    assert open('out/grid.csv').read() == open('out/point.csv').read()
    assert len(open('out/grid.csv').read().strip().split('\n')) == 12
    for k in range(4):
        assert best[k][1:] == ref[k][1:] and abs(best[k][0] - ref[k][0]) < 1e-12


@pytest.mark.parametrize('mode', ['step', 'rollout'])
def test_ac_irl_philox_train_matches_oracle_replay(dev, mode):
    """AC_IRL.train with the in-kernel sampler (rollout with external reward -> reward -> gradient kernel): one
    episode replayed by the oracle on the sampled actions (1-indexed episode, running discount gamma^t, batch-mean
    updates per step or once per episode)."""
    from discrete_mean_field_game_amd import ops
    d, B, gamma = 15, 10, 0.9
    rs = np.random.RandomState(8)
    mat = rs.dirichlet(np.ones(d), size=6)
    np.random.seed(31)
    ac = IRL(theta=8.64, shift=0.0, alpha_scale=1e4, d=d, pi0=mat, demonstrations=[], batch=B, rng='philox', seed=13,
             update_every=mode, precision='f64', verbose=0)
    w0 = ac.w[:, 0].copy(); theta0 = float(np.ravel(ac.theta)[0])
    np.random.seed(32)
    ac.train(max_episodes=1, stop_criteria=-1, gamma=gamma, constant=False, lr_critic=0.1, lr_actor=0.001,
             reward_fn=fake_reward_dev)
    from oracle.philox_ref import start_indices
    idx = start_indices(13, 0, np.arange(B), 6)                     # the device-side start draw of the episode
    pi = mat[idx].astype(np.float32)
    w, theta = w0.copy(), theta0
    sc, sa = O().lr_scales(1, False)
    Gw_acc = np.zeros_like(w); Gt_acc = 0.0
    disc = 1.0
    for t in range(15):
        th = torch.tensor([theta], dtype=torch.float64, device=dev)
        P = ops.sample_dirichlet(torch.as_tensor(pi, device=dev), th, 0.0, 1e4, seed=13, step=t, precision='f64').cpu().numpy()
        pn = O().transition(P, pi).astype(np.float32)
        r = fake_reward_dev(torch.as_tensor(pi, device=dev), torch.as_tensor(P, device=dev)).cpu().numpy().astype(np.float64)
        delta, g, G_w, G_t, _ = O().batched_td_pg(pi, pn, P, r, w, theta, 0.0, disc)
        if mode == 'step':
            w = w + 0.1 * sc * G_w / B
            theta = theta + 0.001 * sa * G_t / B
        else:
            Gw_acc += G_w; Gt_acc += G_t
        disc *= gamma
        pi = pn
    if mode == 'rollout':
        w = w + 0.1 * sc * Gw_acc / (15 * B)
        theta = theta + 0.001 * sa * Gt_acc / (15 * B)
    assert abs(float(np.ravel(ac.theta)[0]) - theta) < 1e-9
    assert np.max(np.abs(ac.w[:, 0] - w)) < 1e-9


@pytest.mark.parametrize('mode', ['step', 'rollout'])
def test_checkpoint_resume_is_bit_identical(dev, mode, tmp_path):
    """state_dict / load_state_dict (theta, w, Philox counter, np.random state): 2 + 2 episodes through a saved file
    equal 4 uninterrupted episodes bit for bit."""
    d, B = 21, 64
    rs = np.random.RandomState(1)
    mat = rs.dirichlet(np.ones(d), size=7)
    np.random.seed(3)
    a = AC(d=d, pi0=mat, batch=B, seed=4, update_every=mode, verbose=0)
    np.random.seed(5)
    a.train(num_episodes=4, gamma=0.9)
    np.random.seed(3)
    b = AC(d=d, pi0=mat, batch=B, seed=4, update_every=mode, verbose=0)
    np.random.seed(5)
    b.train(num_episodes=2, gamma=0.9)
    torch.save(b.state_dict(), str(tmp_path / 'ck.pt'))
    np.random.seed(999)                                   # scramble everything that is not in the checkpoint
    c = AC(d=d, pi0=mat, batch=B, seed=77, update_every=mode, verbose=0)
    c.load_state_dict(torch.load(str(tmp_path / 'ck.pt'), weights_only=True))   # tensors + scalars only
    c.train(num_episodes=2, gamma=0.9, first_episode=2)     # the lr/(episode+1) schedule continues at episode 2
    assert np.array_equal(np.ravel(a.theta), np.ravel(c.theta))
    assert np.array_equal(a.w, c.w)
    assert a._rng_step == c._rng_step


def test_check_finite_raises_like_the_reference_warning_filter(dev):
    """check_finite=True is the stand-in for the reference's warnings-as-errors (mfg_ac2.py:21): a run that blows up
    raises FloatingPointError at the end of the offending episode instead of silently carrying NaNs.  Without it a
    diverged mixed-precision run still cannot go on unnoticed: the sampling kernel that meets the non-finite theta reports
    it through the device status word and the library refuses the next launch (MFG_ERANGE, include/mfg_hip.h)."""
    from discrete_mean_field_game_amd import ops, _lib as L
    d = 21
    rs = np.random.RandomState(0)
    ops.clear_status()
    ac = AC(d=d, pi0=rs.dirichlet(np.ones(d), size=4), batch=32, seed=1, update_every='rollout', check_finite=True, verbose=0)
    ac.train(num_episodes=2)                                   # healthy run: no exception
    with pytest.raises(FloatingPointError):
        ac.train(num_episodes=3, lr_actor=1e300, lr_critic=1e300)
    ops.clear_status()
    ok = AC(d=d, pi0=rs.dirichlet(np.ones(d), size=4), batch=32, seed=1, update_every='rollout', verbose=0)
    with pytest.raises(L.MfgError, match='mfg_clear_status'):  # default: no per-episode check, the status word stops the run
        ok.train(num_episodes=3, lr_actor=1e300, lr_critic=1e300)
    th = np.ravel(ok.theta)[0]
    assert not np.isfinite(th) or abs(th) > 1e100               # (the run was stopped as soon as the diverged theta was used)
    ops.clear_status()
    # strict precision has no range limit and no status report: unchecked, like running with warnings ignored
    f64 = AC(d=d, pi0=rs.dirichlet(np.ones(d), size=4), batch=32, seed=1, update_every='rollout', precision='f64', verbose=0)
    f64.train(num_episodes=4, lr_actor=1e300, lr_critic=1e300)
    assert not (np.isfinite(np.ravel(f64.theta)[0]) and np.all(np.isfinite(f64.w)) and np.abs(f64.w).max() < 1e100)
    assert ops.status() == 0


@pytest.mark.parametrize('mode', ['step', 'rollout'])
def test_ac_irl_checkpoint_resume_is_bit_identical(dev, mode, tmp_path):
    """AC_IRL.state_dict carries everything a resumed IRL run consumes: the reward net + Adam state, the dropout-mask
    call counter of the HIP reward net, Python's `random` state (random.sample in update_reward), torch's generators
    (training-mode dropout), theta_initial, D_samp (list_generated) and the policy FIFO.  reward updates + 2 + 2
    forward episodes through a saved file equal the uninterrupted run bit for bit (dropout active: reg='dropout_l1l2')."""
    import random
    d, B = 15, 48
    rs = np.random.RandomState(2)
    mat = rs.dirichlet(np.ones(d), size=5)
    demos = [[(rs.dirichlet(np.ones(d)), rs.dirichlet(np.ones(d), size=d)) for _ in range(15)] for _ in range(6)]

    def fresh(seed):
        np.random.seed(3); torch.manual_seed(3); random.seed(3)
        return IRL(d=d, pi0=mat, demonstrations=demos, batch=B, num_policies=2, seed=seed, update_every=mode, verbose=0)

    def phase1(ac):
        ac.list_generated = ac.generate_trajectories(6)
        ac.list_eval_gen_transitions = [p for t in ac.list_generated for p in t]
        ac.reward_iteration(max_iterations=4, stop_criteria=-1, iter_check=2)
        ac.train(max_episodes=2, stop_criteria=-1, gamma=0.9)

    def phase2(ac, first):
        ac.reward_iteration(max_iterations=3, stop_criteria=-1, iter_check=2)
        ac.train(max_episodes=2, stop_criteria=-1, gamma=0.9, first_episode=first)   # 2 MORE episodes (3, 4)

    a = fresh(4)
    phase1(a)
    ac_state_after_1 = (a._reward_calls, a._rng_step)
    phase2(a, 2)
    b = fresh(4)
    phase1(b)
    assert (b._reward_calls, b._rng_step) == ac_state_after_1
    torch.save(b.state_dict(), str(tmp_path / 'irl.pt'))
    np.random.seed(999); torch.manual_seed(999); random.seed(999)        # scramble what is not in the checkpoint
    torch.rand(7, device=dev)
    c = IRL(d=d, pi0=mat, demonstrations=demos, batch=B, num_policies=2, seed=77, update_every=mode, verbose=0)
    c.load_state_dict(torch.load(str(tmp_path / 'irl.pt'), weights_only=True))
    c.list_eval_gen_transitions = [p for t in c.list_generated for p in t]
    assert len(c.list_generated) == 6 and np.array_equal(c.list_generated[3][7][1], b.list_generated[3][7][1])
    phase2(c, 2)
    assert np.array_equal(np.ravel(a.theta), np.ravel(c.theta))
    assert np.array_equal(a.w, c.w)
    for pa, pc in zip(a.reward_net.parameters(), c.reward_net.parameters()):
        assert torch.equal(pa, pc)
    assert a._reward_calls == c._reward_calls and a._rng_step == c._rng_step
    assert [float(np.ravel(t)[0]) for t in a.list_policies] == [float(np.ravel(t)[0]) for t in c.list_policies]


def test_ac_irl_outerloop_resume_is_bit_identical(dev, tmp_path):
    """outerloop (ac_irl.py:900-954) stopped after iteration 1 (`final_training=False`), saved, loaded into a fresh object
    and continued with `first_iteration=1` equals the uninterrupted 3-iteration run bit for bit: D_samp, the reward net and
    its optimiser, the policy FIFO, the dropout-mask counter and the host RNG streams all travel in the checkpoint."""
    import random
    d, B = 15, 16
    rs = np.random.RandomState(4)
    mat = rs.dirichlet(np.ones(d), size=5)
    demos = [[(rs.dirichlet(np.ones(d)), rs.dirichlet(np.ones(d), size=d)) for _ in range(15)] for _ in range(6)]
    kw = dict(num_gen_from_policy=2, max_reward_iterations=3, max_forward_episodes=2)

    def fresh(seed):
        np.random.seed(3); torch.manual_seed(3); random.seed(3)
        return IRL(d=d, pi0=mat, demonstrations=demos, batch=B, num_policies=2, seed=seed, verbose=0)

    a = fresh(4)
    a.outerloop(num_iterations=3, final_training=False, **kw)
    b = fresh(4)
    b.outerloop(num_iterations=1, final_training=False, **kw)
    torch.save(b.state_dict(), str(tmp_path / 'outer.pt'))
    np.random.seed(999); torch.manual_seed(999); random.seed(999)
    torch.rand(5, device=dev)
    c = IRL(d=d, pi0=mat, demonstrations=demos, batch=B, num_policies=2, seed=77, verbose=0)
    c.load_state_dict(torch.load(str(tmp_path / 'outer.pt'), weights_only=True))
    assert len(c.list_eval_gen_transitions) == len(b.list_eval_gen_transitions) > 0      # rebuilt on load
    c.outerloop(num_iterations=3, first_iteration=1, final_training=False, **kw)
    assert np.array_equal(np.ravel(a.theta), np.ravel(c.theta)) and np.array_equal(a.w, c.w)
    for pa, pc in zip(a.reward_net.parameters(), c.reward_net.parameters()):
        assert torch.equal(pa, pc)
    assert a.reward_update_count == c.reward_update_count
    assert [float(np.ravel(t)[0]) for t in a.list_policies] == [float(np.ravel(t)[0]) for t in c.list_policies]


def test_start_table_setter_keeps_the_draw_range_and_indices_are_clamped(dev):
    """ADVICE (round 2): assigning `mat_pi0` must move `num_start_samples` with it (mfg_synthetic.train re-reads the table,
    mfg_synthetic.py:438-440), and a stale index can never read outside the table: the kernels clamp it."""
    from discrete_mean_field_game_amd import ops
    d = 21
    rs = np.random.RandomState(0)
    ac = AC(d=d, pi0=rs.dirichlet(np.ones(d), size=9), batch=64, seed=1, update_every='rollout', verbose=0)
    assert ac.num_start_samples == 9
    ac.mat_pi0 = rs.dirichlet(np.ones(d), size=3)
    assert ac.num_start_samples == 3
    np.random.seed(1)
    ac.train(num_episodes=2)                                   # draws stay inside the 3-row table
    tab = torch.as_tensor(ac.mat_pi0.astype(np.float32), device=dev)
    idx = torch.tensor([0, 2, 3, 1000, -5], dtype=torch.int32, device=dev)
    got = ops.gather_start(tab, idx).cpu().numpy()
    want = ac.mat_pi0.astype(np.float32)[[0, 2, 2, 2, 0]]
    assert np.array_equal(got, want)


# ---- f2 / f4 pinned to the unmodified reference (tests/golden/host_io_mfg_ac2.npz) ---------------------------------------
def _eval_setup(z, tmp_path, monkeypatch):
    """Recreate the test files the reference evaluated and make os.listdir return them in the order the reference's
    filesystem did: evaluate() generates one trajectory per file in listdir order, so that order IS the np.random
    consumption order."""
    import discrete_mean_field_game_amd.mfg_ac2 as M
    monkeypatch.chdir(tmp_path)
    os.makedirs('test_normalized_round2'); os.makedirs('out')
    for name, m in zip(z['eval_files'], z['eval_emp']):
        np.savetxt('test_normalized_round2/' + str(name), m, fmt='%.3e', delimiter=' ')
    real = os.listdir
    order = [str(n) for n in z['eval_listdir_order']]
    monkeypatch.setattr(M.os, 'listdir', lambda p: order if str(p).rstrip('/').endswith('test_normalized_round2') else real(p))


def _csv_close(got, want, rel=2.5e-3):
    """Same number of lines / fields, leading (theta, shift, alpha_scale) text identical, '%.3e' fields within one unit
    of the last printed digit (fp32 storage of states and actions vs the reference's fp64)."""
    gl, wl = got.strip().split('\n'), want.strip().split('\n')
    assert len(gl) == len(wl)
    for a, b in zip(gl, wl):
        fa, fb = a.split(','), b.split(',')
        assert len(fa) == len(fb)
        if not fb[0][0].isdigit():
            assert a == b                                              # header line
            continue
        assert fa[:3] == fb[:3]
        for x, y in zip(fa[3:], fb[3:]):
            assert len(x) == len(y) and abs(float(x) - float(y)) <= rel * abs(float(y))


@pytest.mark.parametrize('precision', ['mixed', 'f64'])
def test_evaluate_retraces_reference(dev, tmp_path, monkeypatch, precision):
    """evaluate() with rng='numpy' consumes np.random trajectory-major like mfg_ac2.py:629-633 and reproduces the
    reference's returned tuple and CSV line."""
    z = np.load(os.path.join(G, 'host_io_mfg_ac2.npz'))
    _eval_setup(z, tmp_path, monkeypatch)
    theta, shift, scale = (float(v) for v in z['eval_args'])
    np.random.seed(0)
    ac = AC(d=21, pi0=np.eye(21)[:2], rng='numpy', precision=precision)
    np.random.seed(int(z['eval_seed']))
    res = ac.evaluate(theta=theta, shift=shift, alpha_scale=scale, d=21, outfile='out/e.csv', write_header=1)
    assert np.max(np.abs(np.array(res) - z['eval_result']) / z['eval_result']) < 2e-6
    with open('out/e.csv') as f:
        _csv_close(f.read(), str(z['eval_csv']))


def test_gridsearch_retraces_reference(dev, tmp_path, monkeypatch):
    z = np.load(os.path.join(G, 'host_io_mfg_ac2.npz'))
    _eval_setup(z, tmp_path, monkeypatch)
    np.random.seed(0)
    ac = AC(d=21, pi0=np.eye(21)[:2], rng='numpy')
    np.random.seed(int(z['grid_seed']))
    best = ac.gridsearch(list(z['grid_thetas']), list(z['grid_shifts']), list(z['grid_alphas']),
                         indir='test_normalized_round2', outfile='out/g.csv')
    with open('out/g.csv') as f:
        _csv_close(f.read(), str(z['grid_csv']))
    want = z['grid_best']
    for k in range(4):
        assert abs(best[k][0] - want[k, 0]) < 2e-6 * want[k, 0]
        assert tuple(best[k][1:]) == tuple(want[k, 1:])                # same argmin grid point


def test_train_write_file_csv_schema_matches_reference(dev, tmp_path, monkeypatch):
    """train(write_file=1) appends the reference's three CSV logs (mfg_ac2.py:441-445, :536-539): one line per
    `consecutive` episodes with theta '%.5e', the final pi '%.3e', the windowed average return '%.3e'."""
    z = np.load(os.path.join(G, 'host_io_mfg_ac2.npz'))
    monkeypatch.chdir(tmp_path)
    os.makedirs('results')
    np.random.seed(int(z['log_seed']))
    ac = AC(d=21, pi0=z['log_mat_pi0'], batch=1, rng='numpy', update_every='step', precision='f64')
    assert np.array_equal(ac.w, z['log_w0'])
    ac.train(num_episodes=5, gamma=0.9, constant=0, consecutive=2, write_file=1)
    for name, rel in (('theta', 2e-6), ('pi', 2.5e-3), ('reward', 2.5e-3)):
        with open('results/%s.csv' % name) as f:
            got = f.read()
        want = str(z['log_' + name])
        gl, wl = got.strip().split('\n'), want.strip().split('\n')
        assert len(gl) == len(wl) == 3
        for a, b in zip(gl, wl):
            fa, fb = a.split(','), b.split(',')
            assert len(fa) == len(fb)
            for x, y in zip(fa, fb):
                assert len(x) == len(y) and abs(float(x) - float(y)) <= rel * abs(float(y)) + 1e-12
    assert abs(float(np.ravel(ac.theta)[0]) - float(z['log_theta_final'])) < 2e-7


def test_ac_irl_generate_trajectories_numpy_rng_retraces_reference(dev):
    """AC_IRL.generate_trajectories(rng='numpy') on the GPU vs the unmodified reference's (pi, P) pairs
    (ac_irl.py:735-767: randint then 15 x d gamma vector draws per trajectory, trajectory after trajectory)."""
    z = np.load(os.path.join(G, 'generate_trajectories_ac_irl.npz'))
    np.random.seed(0)
    ac = IRL(theta=float(z['theta']), shift=float(z['shift']), alpha_scale=float(z['alpha_scale']), d=21,
             pi0=z['mat_pi0'], rng='numpy', use_tf=False, num_policies=3)
    np.random.seed(int(z['seed']))
    trajs = ac.generate_trajectories(int(z['n']))
    assert len(trajs) == 2 and all(len(t) == 15 for t in trajs)
    pis = np.array([[p[0] for p in t] for t in trajs]); Ps = np.array([[p[1] for p in t] for t in trajs])
    assert pis.dtype == np.float64 and Ps.shape == (2, 15, 21, 21)
    assert np.allclose(pis, z['pi'], rtol=5e-6, atol=1e-9)
    assert np.allclose(Ps, z['P'], rtol=2e-6, atol=1e-12)
    ac._pi_alpha = None                        # concentrations from the pi passed in (the fixture recomputed alpha per state)
    g = np.array([[ac.calc_gradient_vectorized(Ps[b, t].copy(), pis[b, t]) for t in range(15)] for b in range(2)])
    # mixed-precision score on fp32-rounded (pi, P): late in a trajectory g is a 1e-5-sized sum of cancelling O(1e-2) terms,
    # so the bound is relative to the leading terms, not to g itself
    assert np.allclose(g, z['gradient'], rtol=1e-5, atol=2e-8)


@pytest.mark.parametrize('d,B,T', [(21, 1000, 15), (15, 77, 15), (47, 30, 4), (128, 40, 5)])
@pytest.mark.parametrize('apply', [True, False])
def test_train_rollout_equals_gather_rollout_update_sequence(dev, d, B, T, apply):
    """mfg_train_rollout (start-state gather inside the rollout kernel, update inside the kernel that finishes the sums) gives
    bit for bit what the separate calls give: mfg_gather_start -> mfg_rollout(TD, G) -> mfg_apply_update."""
    from discrete_mean_field_game_amd import ops
    rs = np.random.RandomState(d)
    mat = torch.as_tensor(rs.dirichlet(np.ones(d), size=9).astype(np.float32), device=dev)
    idx = torch.as_tensor(rs.randint(9, size=B).astype(np.int32), device=dev)
    F = ops.num_features(d)
    w0 = rs.rand(F)
    outs = []
    for fused in (True, False):
        theta = torch.tensor([8.86349], dtype=torch.float64, device=dev)
        w = torch.as_tensor(w0.copy(), device=dev)
        G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
        ws = ops.workspace(B * T, d, dev)
        racc = torch.zeros(1, dtype=torch.float64, device=dev)
        bufs = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'pi_last': torch.empty(B, d, device=dev),
                'reward': torch.empty(B, T, device=dev), 'delta': torch.empty(B, T, dtype=torch.float64, device=dev),
                'g': torch.empty(B, T, dtype=torch.float64, device=dev)}
        if fused:
            ops.train_rollout(mat, idx, T, theta, 0.16, 12000.0, w, 0.9, G, ws, bufs, 0.1, 0.001, apply=apply, seed=5,
                              first_step=3, traj_offset=11, reward_acc=racc)
            if not apply:
                ops.apply_update(G, d, 0.1, 0.001, w, theta, racc)
        else:
            pi0 = ops.gather_start(mat, idx)
            ops.rollout(pi0, T, theta, 0.16, 12000.0, w=w, gamma=0.9, seed=5, first_step=3, traj_offset=11, td=True, G=G,
                        ws=ws, out=bufs)
            ops.apply_update(G, d, 0.1, 0.001, w, theta, racc)
        torch.cuda.synchronize()
        outs.append((theta.clone(), w.clone(), G.clone(), racc.clone(), bufs['pi_traj'].clone(), bufs['delta'].clone(),
                     bufs['g'].clone(), bufs['reward'].clone(), bufs['pi_last'].clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert float(outs[0][0]) != 8.86349 and float(outs[0][2][F + 2]) == B * T


# ---- a9 on the device: the per-episode start-state draw (mfg_ac2.py:466) and the native episode loops -----------------------
@pytest.mark.parametrize('num_start', [1, 3, 64, 1000])
def test_draw_start_equals_the_oracle_bit_for_bit(dev, num_start):
    """mfg_draw_start: indices and gathered rows against oracle/philox_ref.start_indices (integer bookkeeping: bit exact),
    for trajectory ids beyond 2^32 and steps near the top of the counter."""
    from discrete_mean_field_game_amd import ops
    from oracle.philox_ref import start_indices
    d = 21
    mat = np.random.RandomState(num_start).dirichlet(np.ones(d), size=num_start).astype(np.float32)
    mat_dev = torch.as_tensor(mat, device=dev)
    for seed, step, off, B in [(0, 0, 0, 1), (2024, 45, 0, 4097), (0x9E3779B97F4A7C15, 4_000_000_000, 2 ** 33 + 5, 1000),
                               (5, 7, 2 ** 47, 13)]:
        idx, pi0 = ops.draw_start(mat_dev, B, seed, step, off, want_idx=True)
        ref = start_indices(seed, step, off + np.arange(B, dtype=np.uint64), num_start)
        assert np.array_equal(idx.cpu().numpy(), ref)
        assert np.array_equal(pi0.cpu().numpy(), mat[ref])
        only_idx, none = ops.draw_start(mat_dev, B, seed, step, off, want_idx=True, want_pi0=False)
        assert none is None and torch.equal(only_idx, idx)


@pytest.mark.parametrize('d,B,T', [(21, 1000, 15), (21, 13, 3), (15, 77, 15), (47, 30, 4), (128, 40, 5), (256, 9, 2), (448, 5, 2)])
def test_in_kernel_start_draw_equals_the_gather_of_the_oracle_indices(dev, d, B, T):
    """mfg_train_rollout(idx = NULL) draws the start rows inside the rollout kernel (packed and wave-per-trajectory
    forms): every output equals, bit for bit, the run that is handed oracle/philox_ref.start_indices as `idx`."""
    from discrete_mean_field_game_amd import ops
    from oracle.philox_ref import start_indices
    rs = np.random.RandomState(d)
    mat_h = rs.dirichlet(np.ones(d), size=9).astype(np.float32)
    mat = torch.as_tensor(mat_h, device=dev)
    seed, first_step, off = 5, 3 * T, 2 ** 32 + 11
    ref = start_indices(seed, first_step, off + np.arange(B, dtype=np.uint64), 9)
    idx = torch.as_tensor(ref.astype(np.int32), device=dev)
    F = ops.num_features(d)
    w0 = rs.rand(F)
    outs = []
    for drawn in (True, False):
        theta = torch.tensor([8.86349], dtype=torch.float64, device=dev)
        w = torch.as_tensor(w0.copy(), device=dev)
        G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
        ws = ops.workspace(B * T, d, dev)
        bufs = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'pi_last': torch.empty(B, d, device=dev),
                'reward': torch.empty(B, T, device=dev), 'delta': torch.empty(B, T, dtype=torch.float64, device=dev),
                'g': torch.empty(B, T, dtype=torch.float64, device=dev)}
        ops.train_rollout(mat, None if drawn else idx, T, theta, 0.16, 12000.0, w, 0.9, G, ws, bufs, 0.1, 0.001, apply=True,
                          seed=seed, first_step=first_step, traj_offset=off)
        torch.cuda.synchronize()
        outs.append((theta.clone(), w.clone(), G.clone(), bufs['pi_traj'].clone(), bufs['delta'].clone(), bufs['g'].clone(),
                     bufs['reward'].clone(), bufs['pi_last'].clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    assert np.array_equal(outs[0][3][:, 0].cpu().numpy(), mat_h[ref])


@pytest.mark.parametrize('mode,d,B', [('rollout', 21, 300), ('step', 21, 300), ('rollout', 15, 64), ('rollout', 128, 12), ('step', 47, 20)])
@pytest.mark.parametrize('constant', [0, 1])
def test_native_multi_episode_loop_equals_the_per_episode_loop(dev, mode, d, B, constant):
    """mfg_train_rollouts / mfg_train_episodes (ALL episodes up to the next report issued by one native call: device-side
    start draw, learning-rate schedule evaluated in C, episode loop without the interpreter) against the per-episode
    Python loop of the same kernels (a recorded trace forces it): theta, w, final states, the Philox counter and the
    per-report reward averages agree bit for bit -- including the libm `log(log(episode + 20))` of the actor schedule."""
    rs = np.random.RandomState(d)
    mat = rs.dirichlet(np.ones(d), size=9)
    runs = []
    for native in (True, False):
        np.random.seed(21)
        ac = AC(d=d, pi0=mat, batch=B, rng='philox', seed=5, update_every=mode, verbose=0)
        if not native:
            ac.trace = []
        logs = []
        ac.train_log = lambda vector, filename, fmt, logs=logs: logs.append((filename, np.array(vector, dtype=np.float64).copy()))
        ac.train(num_episodes=8, gamma=0.95, constant=constant, consecutive=3, write_file=1, first_episode=2)
        runs.append((np.ravel(ac.theta).copy(), ac.w.copy(), ac._last_pi.cpu().numpy().copy(), ac._rng_step, logs))
    assert runs[0][3] == runs[1][3] == 8 * 15
    for k in range(3):
        assert np.array_equal(runs[0][k], runs[1][k])
    assert len(runs[0][4]) == len(runs[1][4]) == 9                   # reports after episodes 0, 3, 6: theta, pi, reward each
    for (fa, va), (fb, vb) in zip(runs[0][4], runs[1][4]):
        assert fa == fb and np.array_equal(va, vb)


@pytest.mark.parametrize('d,B,T', [(21, 1000, 15), (15, 77, 15), (47, 30, 4), (128, 40, 5)])
def test_deferred_update_chain_equals_update_launches(dev, d, B, T):
    """mfg_train_rollout_deferred: the multi-rank cycle rollout -> sums -> [all-reduce] with the update of episode e applied
    inside episode e+1's rollout kernel (weight staging; out of place at d > 64) gives bit for bit what
    rollout -> sums -> [all-reduce] -> mfg_apply_update gives: parameters after every episode, the per-episode reward
    accumulators and all outputs of the last rollout."""
    from discrete_mean_field_game_amd import ops
    rs = np.random.RandomState(d)
    mat = torch.as_tensor(rs.dirichlet(np.ones(d), size=9).astype(np.float32), device=dev)
    F = ops.num_features(d)
    w0 = rs.rand(F)
    E = 4
    lrs = [(0.1 / (e + 1), 0.001 / (e + 2)) for e in range(E)]
    outs = []
    for deferred in (True, False):
        theta = torch.tensor([8.86349], dtype=torch.float64, device=dev)
        w = torch.as_tensor(w0.copy(), device=dev)
        theta2, w2 = torch.empty_like(theta), torch.empty_like(w)
        G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
        ws = ops.workspace(B * T, d, dev)
        racc = torch.zeros(E, dtype=torch.float64, device=dev)
        bufs = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'pi_last': torch.empty(B, d, device=dev),
                'reward': torch.empty(B, T, device=dev), 'delta': torch.empty(B, T, dtype=torch.float64, device=dev),
                'g': torch.empty(B, T, dtype=torch.float64, device=dev)}
        hist = []
        pending = None
        for e in range(E):
            if deferred:
                ops.train_rollout_deferred(mat, None, T, theta, w, pending, theta2, w2, 0.16, 12000.0, 0.9, G, ws, bufs, seed=5,
                                           first_step=e * T, traj_offset=11)
                if pending is not None:
                    theta, theta2, w, w2 = theta2, theta, w2, w
                    hist.append((theta.clone(), w.clone()))             # parameters after update e-1
                pending = (G, lrs[e][0], lrs[e][1], racc.data_ptr() + 8 * e)
            else:
                ops.train_rollout(mat, None, T, theta, 0.16, 12000.0, w, 0.9, G, ws, bufs, apply=False, seed=5,
                                  first_step=e * T, traj_offset=11)
                ops.apply_update(G, d, lrs[e][0], lrs[e][1], w, theta, racc.data_ptr() + 8 * e)
                hist.append((theta.clone(), w.clone()))
        if deferred:
            ops.apply_update(G, d, lrs[-1][0], lrs[-1][1], w, theta, racc.data_ptr() + 8 * (E - 1))
            hist.append((theta.clone(), w.clone()))
        torch.cuda.synchronize()
        outs.append((hist, racc.clone(), [bufs[k].clone() for k in ('pi_traj', 'reward', 'delta', 'g', 'pi_last')], G.clone()))
    assert len(outs[0][0]) == len(outs[1][0]) == E
    for (ta, wa), (tb, wb) in zip(outs[0][0], outs[1][0]):
        assert torch.equal(ta, tb) and torch.equal(wa, wb)
    assert torch.equal(outs[0][1], outs[1][1]) and bool((outs[0][1] != 0).all())
    for x, y in zip(outs[0][2], outs[1][2]):
        assert torch.equal(x, y)
    assert torch.equal(outs[0][3], outs[1][3])
    assert float(outs[0][0][-1][0]) != 8.86349


def test_native_loops_edge_cases_and_error_paths(dev):
    """mfg_train_rollouts / mfg_train_episodes / mfg_draw_start / mfg_train_rollout_deferred: zero episodes and an empty batch
    are no-ops, even and odd episode lengths leave the final states where the per-episode calls leave them, bad arguments
    are refused with an error code before anything is launched."""
    from discrete_mean_field_game_amd import ops, _lib as L
    lib = L.lib()
    d, B = 21, 50
    rs = np.random.RandomState(4)
    mat = torch.as_tensor(rs.dirichlet(np.ones(d), size=5).astype(np.float32), device=dev)
    F = ops.num_features(d)

    def fresh(T):
        return dict(theta=torch.tensor([8.86349], dtype=torch.float64, device=dev), w=torch.as_tensor(rs.rand(F) * 0 + 0.3, device=dev),
                    G=torch.zeros(F + 3, dtype=torch.float64, device=dev), ws=ops.workspace(B * T, d, dev),
                    bufs={'pi_traj': torch.empty(B, T + 1, d, device=dev), 'pi_last': torch.empty(B, d, device=dev),
                          'reward': torch.empty(B, T, device=dev), 'delta': torch.empty(B, T, dtype=torch.float64, device=dev),
                          'g': torch.empty(B, T, dtype=torch.float64, device=dev)})
    # zero episodes: nothing moves
    z = fresh(15)
    ops.train_rollouts(mat, 15, 0, 0, False, z['theta'], 0.16, 12000.0, z['w'], 1.0, z['G'], z['ws'], z['bufs'], 0.1, 0.001, seed=1)
    torch.cuda.synchronize()
    assert float(z['theta'][0]) == 8.86349 and float(z['G'].abs().sum()) == 0.0
    # step-mode episodes of even and odd length: final states = those of per-episode mfg_train_episode calls
    for T in (1, 2, 5, 6):
        a, b2 = fresh(T), fresh(T)
        eb = ops.episode_buffers(B, d, dev)
        pi_a = torch.empty(B, d, device=dev)
        ops.train_episodes(mat, pi_a, T, 3, 1, False, a['theta'], 0.16, 12000.0, a['w'], 0.9, 0.1, 0.001, a['G'], a['ws'], eb, seed=7,
                           first_step=40)
        from discrete_mean_field_game_amd.parallel import lr_scales
        pi_b = None
        for k in range(3):
            _, pi_b = ops.draw_start(mat, B, 7, 40 + k * T)
            sc, sa = lr_scales(1 + k, False)
            ops.train_episode(pi_b, T, b2['theta'], 0.16, 12000.0, b2['w'], 0.9, 0.1 * sc, 0.001 * sa, b2['G'], b2['ws'],
                              ops.episode_buffers(B, d, dev), seed=7, first_step=40 + k * T)
        torch.cuda.synchronize()
        assert torch.equal(a['theta'], b2['theta']) and torch.equal(a['w'], b2['w']) and torch.equal(pi_a, pi_b), T
    # error paths: empty / oversized table, nothing to write, aliasing parameter sets of the deferred update, wrapped step counter
    idx = torch.zeros(B, dtype=torch.int32, device=dev)
    st = None
    assert lib.mfg_draw_start(mat.data_ptr(), 0, B, d, 1, 0, 0, idx.data_ptr(), None, st) == -1
    assert lib.mfg_draw_start(mat.data_ptr(), 1 << 31, B, d, 1, 0, 0, idx.data_ptr(), None, st) == -1
    assert lib.mfg_draw_start(mat.data_ptr(), 5, B, d, 1, 0, 0, None, None, st) == -1
    assert lib.mfg_draw_start(None, 5, B, d, 1, 0, 0, None, z['bufs']['pi_last'].data_ptr(), st) == -1
    assert lib.mfg_draw_start(mat.data_ptr(), 5, 0, d, 1, 0, 0, idx.data_ptr(), None, st) == 0          # B = 0: no-op
    zb = z['bufs']
    args = [mat.data_ptr(), 5, None, B, d, 15, z['theta'].data_ptr(), z['w'].data_ptr(), z['G'].data_ptr(), 0.1, 0.001, None,
            z['theta'].data_ptr(), z['w'].data_ptr(),                                                       # outputs alias the inputs
            0.16, 12000.0, 1.0, 0, 1, 0, 0, 0, zb['pi_traj'].data_ptr(), zb['pi_last'].data_ptr(), zb['reward'].data_ptr(),
            zb['delta'].data_ptr(), zb['g'].data_ptr(), z['G'].data_ptr(), z['ws'].data_ptr(), z['ws'].numel() * 8, st]
    assert lib.mfg_train_rollout_deferred(*args) == -1 and b'separate output' in lib.mfg_last_error()
    rc = lib.mfg_train_rollouts(mat.data_ptr(), 5, B, d, 15, 10, 0, 0, z['theta'].data_ptr(), 0.16, 12000.0, z['w'].data_ptr(), 1.0, 0, 1,
                                0xFFFFFFF0, 0, 0, 0.1, 0.001, zb['pi_traj'].data_ptr(), zb['pi_last'].data_ptr(),
                                zb['reward'].data_ptr(), zb['delta'].data_ptr(), zb['g'].data_ptr(), z['G'].data_ptr(), None,
                                z['ws'].data_ptr(), z['ws'].numel() * 8, st)
    assert rc == -1 and b'wrap' in lib.mfg_last_error()
    torch.cuda.synchronize()
    assert float(z['theta'][0]) == 8.86349                                                                 # nothing was launched


def test_batched_philox_training_uses_no_host_rng(dev):
    """Batched runs draw their start states on the device: train() neither consumes nor depends on np.random (the
    reference's `np.random.randint` at mfg_ac2.py:466 is kept for batch 1 and rng='numpy' only), and the states an episode
    starts from are the rows oracle/philox_ref.start_indices names."""
    from oracle.philox_ref import start_indices
    d, B = 21, 200
    mat = np.random.RandomState(0).dirichlet(np.ones(d), size=16)
    outs = []
    for host_seed in (1, 2):
        np.random.seed(5)
        ac = AC(d=d, pi0=mat, batch=B, seed=9, update_every='rollout')
        np.random.seed(host_seed)
        before = np.random.get_state()[1].copy()
        ac.train(num_episodes=3)
        assert np.array_equal(before, np.random.get_state()[1])     # the stream was not touched
        outs.append((float(ac.theta[0]), ac.w.copy()))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])
    # per-step mode, per-episode Python path (trace): states of episode e start from start_indices(seed, 15 e, b)
    np.random.seed(5)
    ac = AC(d=d, pi0=mat, batch=B, seed=9, update_every='step')
    ac.trace = []
    from discrete_mean_field_game_amd import ops
    seen = []
    real = ops.draw_start
    try:
        ops.draw_start = lambda *a, **k: seen.append((a[2], a[3])) or real(*a, **k)
        ac.train(num_episodes=2)
    finally:
        ops.draw_start = real
    assert seen == [(9, 0), (9, 15)]
    _, pi0 = real(ac._mat_pi0_dev, B, 9, 15, 0)
    assert np.array_equal(pi0.cpu().numpy(), mat.astype(np.float32)[start_indices(9, 15, np.arange(B), 16)])


def test_batch_one_and_numpy_rng_keep_the_reference_host_draw(dev):
    """batch = 1 (and rng = 'numpy' at any batch) still take `np.random.randint(num_start_samples)` per episode, the
    reference's draw (mfg_ac2.py:466): one scalar draw per episode at batch 1."""
    d = 21
    mat = np.random.RandomState(0).dirichlet(np.ones(d), size=16)
    np.random.seed(5)
    ac = AC(d=d, pi0=mat, batch=1, seed=9, rng='philox', update_every='step')
    np.random.seed(77)
    ac.train(num_episodes=4)
    after = np.random.get_state()
    np.random.seed(77)
    for _ in range(4):
        np.random.randint(16)
    assert np.array_equal(after[1], np.random.get_state()[1]) and after[2] == np.random.get_state()[2]


@pytest.mark.parametrize('d,B', [(21, 1), (21, 2), (21, 3), (21, 4), (21, 100), (15, 1), (15, 5), (15, 1001)])
def test_given_P_wave_kernel_ragged_tiles(dev, d, B):
    """k_step_wave (wave-private tiles, 16-byte loads of the enclosing aligned window) at batch sizes that leave ragged
    last tiles and last 16-byte words, against the oracle; the reward-less call returns the same states."""
    from discrete_mean_field_game_amd import ops
    rs = np.random.RandomState(B)
    pi = rs.dirichlet(np.ones(d), size=B).astype(np.float32)
    P = rs.dirichlet(np.ones(d), size=(B, d)).astype(np.float32)
    pn, r = ops.step_given_P(torch.as_tensor(pi, device=dev), torch.as_tensor(P, device=dev))
    assert np.array_equal(pn.cpu().numpy(), O().transition(P.astype(np.float64), pi.astype(np.float64)).astype(np.float32))
    ref = O().calc_reward(P.astype(np.float64), pi.astype(np.float64))
    assert np.max(np.abs(r.cpu().numpy() - ref) / np.maximum(np.abs(ref), 1e-6)) < 1e-6
    pn2, none = ops.step_given_P(torch.as_tensor(pi, device=dev), torch.as_tensor(P, device=dev), want_reward=False)
    assert none is None and torch.equal(pn2, pn)


def test_contexts_isolate_the_status_word_between_instances(dev):
    """SURVEY.md 8b / VERDICT r4 item 4: the sticky status word belongs to a context (mfg_ctx_t); every model instance owns one
    and binds it in its public methods.  A diverged mixed-precision run poisons ITS word: it is refused from then on
    (MFG_ERANGE), while a second instance on the same device -- and a caller that never binds a context (the default one) --
    keep sampling.  Clearing is per context too."""
    import ctypes as C
    from discrete_mean_field_game_amd import ops, _lib as L
    d = 21
    rs = np.random.RandomState(0)
    mat = rs.dirichlet(np.ones(d), size=4)
    a = AC(d=d, pi0=mat, batch=32, seed=1, update_every='rollout', verbose=0)
    b = AC(d=d, pi0=mat, batch=32, seed=2, update_every='rollout', verbose=0)
    assert a._ctx._ptr != b._ctx._ptr
    with pytest.raises(L.MfgError, match='mfg_clear_status'):
        a.train(num_episodes=3, lr_actor=1e300, lr_critic=1e300)          # diverges: theta leaves the mixed-precision range
    assert a.status() == L.STATUS_MIXED_RANGE and b.status() == 0
    b.train(num_episodes=2)                                               # the other instance is not stopped ...
    assert np.isfinite(np.ravel(b.theta)[0]) and b.status() == 0
    P = b.sample_action(mat[0])
    assert np.all(np.isfinite(P))
    a.theta = 8.86349                                                     # ... while the poisoned one stays refused until cleared
    with pytest.raises(L.MfgError):
        a.sample_action(mat[0])
    # a caller without a context: the device's default word, untouched by either instance
    ops.Context.unbind()
    assert L.lib().mfg_ctx_current() is None and ops.status() == 0
    th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
    pi = torch.as_tensor(mat.astype(np.float32), device=dev)
    assert torch.isfinite(ops.sample_dirichlet(pi, th, 0.16, 12000.0, seed=3)).all()
    a.clear_status()
    a.w = a.init_w(d)                                                     # (the diverged critic weights too)
    assert a.status() == 0 and np.all(np.isfinite(a.sample_action(mat[0])))
    assert L.lib().mfg_ctx_current() is None         # the method bound its instance's context and put the caller's back
    # a context of its own for plain ops.* callers; destroying the bound context unbinds it
    c = ops.Context(dev).bind()
    assert L.lib().mfg_ctx_current() == c._ptr and c.status() == 0
    assert np.all(np.isfinite(b.sample_action(mat[0]))) and L.lib().mfg_ctx_current() == c._ptr   # (restored after a method)
    c.close()
    assert L.lib().mfg_ctx_current() is None
    assert L.lib().mfg_abi_version() >= 17


def test_context_destroyed_on_another_thread_leaves_no_dangling_binding(dev):
    """a context may be destroyed by another thread than the one it is bound on (a garbage collector drops the last
    reference wherever it runs).  The binding thread must then fall back to the device's default word at its next entry
    point instead of dereferencing the freed object; binding / querying a destroyed context is refused; a younger context at
    the same address is not mistaken for the old one."""
    import threading
    from discrete_mean_field_game_amd import ops, _lib as L
    lib = L.lib()
    ops.Context.unbind()
    c = ops.Context(dev).bind()
    ptr = c._ptr
    assert lib.mfg_ctx_current() == ptr
    t = threading.Thread(target=c.close)            # destroyed elsewhere while still bound HERE
    t.start(); t.join()
    assert lib.mfg_ctx_current() is None            # the stale binding is noticed, not followed
    assert ops.status() == 0                        # entry points run on the default word
    th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
    pi = torch.full((2, 21), 1.0 / 21, device=dev)
    assert torch.isfinite(ops.sample_dirichlet(pi, th, 0.16, 12000.0, seed=3)).all()
    assert lib.mfg_ctx_bind(ptr) == L.lib().mfg_ctx_bind(ptr) != 0          # a destroyed context cannot be bound ...
    assert lib.mfg_ctx_status(ptr, None) != 0 and lib.mfg_ctx_destroy(ptr) != 0   # ... queried, or destroyed twice
    # many create / destroy cycles: addresses get reused, a binding made to an OLD object never resolves to a younger one
    olds = []
    for _ in range(8):
        k = ops.Context(dev).bind()
        olds.append(k._ptr)
        t = threading.Thread(target=k.close)
        t.start(); t.join()
        fresh = ops.Context(dev)                    # may land on the address just freed
        assert lib.mfg_ctx_current() is None        # this thread's binding belonged to the destroyed one
        fresh.close()
    # an instance dropped by another thread while a second instance keeps working here
    rs = np.random.RandomState(0)
    mat = rs.dirichlet(np.ones(21), size=4)
    a = AC(d=21, pi0=mat, batch=8, seed=1, update_every='rollout', verbose=0)
    b = AC(d=21, pi0=mat, batch=8, seed=2, update_every='rollout', verbose=0)
    a._ctx.bind()
    box = [b]

    def drop():
        box.pop()._ctx.close()
    del b
    t = threading.Thread(target=drop)
    t.start(); t.join()
    a.train(num_episodes=2)
    assert np.isfinite(np.ravel(a.theta)[0]) and a.status() == 0
    ops.Context.unbind()
