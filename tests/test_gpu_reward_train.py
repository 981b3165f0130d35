"""Reward learning on the device (reference ac_irl.py:382-418 loss + Adam, :804-846 update_reward):
mfg_reward_net_train_step against the fp64 analytic gradient of oracle/reward_net_oracle.py (which tests/test_reward_learning.py
checks against PyTorch autograd and central differences on the CPU) and against networks.RewardNet autograd.
PARITY UNPINNED vs TensorFlow (TF 1.x cannot run here): what these tests pin is the restated graph."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import reward_net_oracle as RO


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _net(d, reg, n3, n4, dev, k1=5, f2=2, k2=3, seed=0):
    from discrete_mean_field_game_amd.networks import RewardNet
    torch.manual_seed(seed)
    net = RewardNet(d=d, reg=reg, f1=1, k1=k1, f2=f2, k2=k2, n_fc3=n3, n_fc4=n4)
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))          # biases away from zero, some negative weights
    return net.to(dev)


def _stores(d, n_demo, n_gen, dev, rs, T=15):
    """Two stores whose physical rows differ from the logical order (pushes with drops in between)."""
    from discrete_mean_field_game_amd.reward_learning import TrajectoryStore
    out = []
    for n in (n_demo, n_gen):
        st = TrajectoryStore(d, T, dev)
        junk = torch.as_tensor(rs.dirichlet(np.ones(d), size=(3, T)), dtype=torch.float32)
        junkP = torch.as_tensor(rs.dirichlet(np.ones(d), size=(3, T, d)), dtype=torch.float32)
        st.push(junk, junkP)
        s = torch.as_tensor(rs.dirichlet(np.ones(d) * 0.7, size=(n, T)), dtype=torch.float32)
        a = torch.as_tensor(rs.dirichlet(np.ones(d) * 0.5, size=(n, T, d)), dtype=torch.float32)
        st.push(s, a, drop=2)                            # logical: [junk2, new...]; rows of `new` reuse freed rows
        out.append(st)
    return out


def _batch_np(store, logical):
    s, a = store.gather(logical)
    d = store.d
    return s.reshape(-1, d).cpu().numpy().astype(np.float64), a.reshape(-1, d, d).cpu().numpy().astype(np.float64)


def _grad_scale(prm, ds, da, gs, ga, n_div, n_traj, masks, l1l2):
    """Per-parameter magnitude of the terms the batch gradient sums: the gradient with |dL/dr_n| as sample weights.  The
    demonstration (-) and generated (+) halves cancel to a small net value (e.g. out_b: 15 - 15 -> 0.03), and an fp32
    result can only be accurate relative to what was summed; tolerances below are 1e-5 of max(|net|, this scale)."""
    nd = ds.shape[0]
    state, action = np.concatenate([ds, gs], 0), np.concatenate([da, ga], 0)
    r, cache = RO.forward_cache(prm, state, action, masks)
    S = r[nd:].reshape(n_traj, -1).sum(1)
    soft = np.exp(S - S.max()); soft /= soft.sum()
    dr = np.concatenate([np.full((nd, 1), 1.0 / n_div), np.repeat(soft, r[nd:].shape[0] // n_traj)[:, None]], 0)
    cache_abs = dict(cache)
    g = RO.backward(prm, cache_abs, dr)
    return np.abs(RO.flatten_like_kernel(g))


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))


GEOMS = [  # d, n3, n4, k1, f2, k2
    (21, 8, 4, 5, 2, 3), (15, 8, 4, 5, 2, 3), (32, 4, 4, 5, 2, 3), (4, 3, 2, 5, 2, 3), (21, 16, 8, 5, 2, 3),
    (9, 5, 3, 3, 1, 5), (12, 32, 32, 7, 2, 7), (15, 6, 4, 1, 2, 1)]


@pytest.mark.parametrize('reg', ['none', 'dropout_l1l2', 'l1l2', 'dropout'])
@pytest.mark.parametrize('d,n3,n4,k1,f2,k2', GEOMS)
def test_gradient_and_loss_match_the_fp64_oracle(dev, reg, d, n3, n4, k1, f2, k2):
    from discrete_mean_field_game_amd.reward_learning import RewardTrainer
    rs = np.random.RandomState(d * 100 + n3)
    net = _net(d, reg, n3, n4, dev, k1, f2, k2)
    demo, gen = _stores(d, 7, 9, dev, rs)
    tr = RewardTrainer(net, 1e-4)
    before = tr.flat.clone()
    demo_idx, gen_idx = [4, 1, 6, 2, 5], [7, 0, 3, 8, 1]
    seed = 0xABCDEF0123 + d
    tr.step(demo, [demo.rows[i] for i in demo_idx], gen, [gen.rows[i] for i in gen_idx], 5, seed, grad_only=True)
    torch.cuda.synchronize()
    assert torch.equal(before, tr.flat) and tr.step_count == 0                      # gradient only: nothing applied
    ds, da = _batch_np(demo, demo_idx)
    gs, ga = _batch_np(gen, gen_idx)
    prm = RO.params_from_torch(net)
    masks = RO.dropout_masks(net.keep_prob, seed, 0, 150, n3, n4) if net.use_dropout else None
    (loss, first, second, regv), g, r = RO.irl_loss_and_grad(prm, ds, da, gs, ga, 5, 5, l1l2=net.use_l1l2, masks=masks)
    ref = RO.flatten_like_kernel(g)
    scale = _grad_scale(prm, ds, da, gs, ga, 5, 5, masks, net.use_l1l2)
    got = tr.grad.cpu().numpy().astype(np.float64)
    offs = np.cumsum([0] + [p.numel() for p in net.parameters()])
    for k in range(10):                                     # per tensor: 1e-5 of its largest entry / of the summed magnitude
        a, b, sc = got[offs[k]:offs[k + 1]], ref[offs[k]:offs[k + 1]], scale[offs[k]:offs[k + 1]]
        assert np.max(np.abs(a - b)) <= 1e-5 * max(np.max(np.abs(b)), np.max(sc), 1e-3), \
            (RO.FLAT_ORDER[k], np.max(np.abs(a - b)), np.max(np.abs(b)), np.max(sc))
    st = tr.stats.cpu().numpy().astype(np.float64)
    assert abs(st[0] - loss) <= 2e-6 * max(1.0, abs(loss)) and abs(st[1] - first) <= 1e-5 and abs(st[2] - second) <= 1e-5
    assert abs(st[3] - regv) <= 2e-6 * max(1.0, regv)


@pytest.mark.parametrize('reg', ['none', 'l1l2'])
@pytest.mark.parametrize('d', [15, 21])
def test_gradient_matches_torch_autograd_of_the_module(dev, reg, d):
    """Same batch through networks.RewardNet + maxent_irl_loss + autograd in fp32 (the round-4 update_reward)."""
    from discrete_mean_field_game_amd.networks import maxent_irl_loss
    from discrete_mean_field_game_amd.reward_learning import RewardTrainer
    rs = np.random.RandomState(3)
    net = _net(d, reg, 8, 4, dev)
    demo, gen = _stores(d, 6, 6, dev, rs)
    tr = RewardTrainer(net, 1e-4)
    di, gi = [0, 2, 3, 5, 1], [5, 4, 0, 2, 3]
    tr.step(demo, [demo.rows[i] for i in di], gen, [gen.rows[i] for i in gi], 5, 1, grad_only=True)
    ds, da = demo.gather(di)
    gs, ga = gen.gather(gi)
    loss, _, _ = maxent_irl_loss(net(ds.reshape(-1, d), da.reshape(-1, d, d)), net(gs.reshape(-1, d), ga.reshape(-1, d, d)), 5, 5,
                                 net.regularization() if net.use_l1l2 else None)
    grads = torch.autograd.grad(loss, list(net.parameters()))
    ref = torch.cat([g.reshape(-1) for g in grads]).double().cpu().numpy()
    got = tr.grad.double().cpu().numpy()
    offs = np.cumsum([0] + [p.numel() for p in net.parameters()])
    prm = RO.params_from_torch(net)
    scale = _grad_scale(prm, *_batch_np(demo, di), *_batch_np(gen, gi), 5, 5, None, net.use_l1l2)
    for k in range(10):
        a, b, sc = got[offs[k]:offs[k + 1]], ref[offs[k]:offs[k + 1]], scale[offs[k]:offs[k + 1]]
        assert np.max(np.abs(a - b)) <= 1e-5 * max(np.max(np.abs(b)), np.max(sc), 1e-3), RO.FLAT_ORDER[k]
    assert abs(float(tr.stats[0]) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))


@pytest.mark.parametrize('reg', ['none', 'dropout_l1l2'])
def test_adam_updates_follow_the_oracle_over_several_steps(dev, reg):
    """Five consecutive update steps (fresh batch and masks each) against oracle gradient + tf.train.AdamOptimizer in fp64.
    Adam's first steps are sign-like (m / sqrt(v) = +-1), which amplifies a relative gradient error where |g| ~ 0: entries
    whose oracle gradient is below 1e-6 of the tensor's largest are compared at 2 lr, the rest at 1e-5 relative."""
    from discrete_mean_field_game_amd.reward_learning import RewardTrainer
    d, lr = 21, 1e-3
    rs = np.random.RandomState(11)
    net = _net(d, reg, 8, 4, dev)
    demo, gen = _stores(d, 8, 8, dev, rs)
    tr = RewardTrainer(net, lr)
    p = tr.flat.double().cpu().numpy()
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    pyr = random.Random(5)
    for step in range(1, 6):
        di, gi = pyr.sample(range(8), 5), pyr.sample(range(8), 5)
        seed = 1000 + step
        # oracle first (weights before the update)
        prm = RO.params_from_torch(net)
        ds, da = _batch_np(demo, di)
        gs, ga = _batch_np(gen, gi)
        masks = RO.dropout_masks(net.keep_prob, seed, 0, 150, 8, 4) if net.use_dropout else None
        _, g, _ = RO.irl_loss_and_grad(prm, ds, da, gs, ga, 5, 5, l1l2=net.use_l1l2, masks=masks)
        gflat = RO.flatten_like_kernel(g)
        p_dev = tr.flat.double().cpu().numpy()
        p_ref, m, v = RO.adam_tf(p_dev, gflat, m, v, step, lr=lr)
        tr.step(demo, [demo.rows[i] for i in di], gen, [gen.rows[i] for i in gi], 5, seed)
        got = tr.flat.double().cpu().numpy()
        assert tr.step_count == step
        big = np.abs(gflat) > 1e-6 * np.abs(gflat).max()
        assert np.max(np.abs(got - p_ref)[big]) <= 1e-5 * np.max(np.abs(p_ref)) + 2e-2 * lr, step
        assert np.max(np.abs(got - p_ref)) <= 2.5 * lr
        # the moments follow too (they carry the history): compare where the gradient is resolved
        assert _rel(tr.m.double().cpu().numpy()[big], m[big]) <= 1e-4
        m = tr.m.double().cpu().numpy()                 # continue from the device state so errors do not compound in the check
        v = tr.v.double().cpu().numpy()
    # module parameters are views of the flat buffer: the forward kernel / state_dict see the update
    assert torch.equal(torch.cat([q.reshape(-1) for q in net.parameters()]), tr.flat)


def test_adam_entry_point_and_split_update_equal_the_fused_one(dev):
    """Multi-GPU form: gradient-only launch + mfg_reward_net_adam == the fused update, bit for bit."""
    from discrete_mean_field_game_amd.reward_learning import RewardTrainer
    rs = np.random.RandomState(2)
    d = 15
    demo, gen = _stores(d, 5, 5, dev, rs)
    a = RewardTrainer(_net(d, 'dropout_l1l2', 8, 4, dev), 1e-4)
    b = RewardTrainer(_net(d, 'dropout_l1l2', 8, 4, dev), 1e-4)
    assert torch.equal(a.flat, b.flat)
    rows = list(range(5))
    for step in range(3):
        a.step(demo, [demo.rows[i] for i in rows], gen, [gen.rows[i] for i in rows], 5, 77 + step)
        b.step(demo, [demo.rows[i] for i in rows], gen, [gen.rows[i] for i in rows], 5, 77 + step, grad_only=True)
        b.apply_grad()
    assert torch.equal(a.flat, b.flat) and torch.equal(a.m, b.m) and torch.equal(a.v, b.v) and a.step_count == b.step_count == 3


def test_error_paths(dev):
    from discrete_mean_field_game_amd import _lib as L
    from discrete_mean_field_game_amd.reward_learning import RewardTrainer
    rs = np.random.RandomState(2)
    d = 15
    demo, gen = _stores(d, 5, 5, dev, rs)
    tr = RewardTrainer(_net(d, 'none', 8, 4, dev), 1e-4)
    with pytest.raises(L.MfgError):                       # too many trajectories for the by-value row list
        tr.step(demo, [0] * 65, gen, [0], 5, 1)
    with pytest.raises(L.MfgError):                       # empty batch
        tr.step(demo, [], gen, [], 5, 1)
    with pytest.raises(L.MfgError, match='batch too large'):   # 64 + 64 trajectories x 15 with n_fc3 = 8: 69 KB of coefficients
        tr.step(demo, [0] * 64, gen, [0] * 64, 5, 1)
    tr._ws = torch.empty(8, dtype=torch.float32, device=dev)
    tr._workspace = lambda n: tr._ws
    with pytest.raises(L.MfgError):                       # workspace too small
        tr.step(demo, [0], gen, [0], 5, 1)


def _irl(dev, d=15, B=32, reg='dropout_l1l2', seed=5, n_demo=7, **kw):
    from discrete_mean_field_game_amd.ac_irl import AC_IRL
    rs = np.random.RandomState(4)
    mat = rs.dirichlet(np.ones(d), size=6)
    demos = [[(rs.dirichlet(np.ones(d)), rs.dirichlet(np.ones(d), size=d)) for _ in range(15)] for _ in range(n_demo)]
    np.random.seed(1); torch.manual_seed(1)
    return AC_IRL(d=d, pi0=mat, demonstrations=demos, batch=B, num_policies=2, seed=seed, reg=reg, verbose=0, device=dev, **kw), demos


def test_update_reward_draws_the_reference_batches_and_trains_on_the_device(dev):
    """AC_IRL.update_reward: the batch is the one `random.sample(self.list_demonstrations, 5)` / `(self.list_generated, 5)`
    would pick (ac_irl.py:814-829), the update equals the oracle's on that batch, and the host `random` stream advances
    exactly like the reference's two sample() calls."""
    ac, demos = _irl(dev, reg='l1l2')
    ac.list_generated = ac.generate_trajectories(8)
    gens = ac.list_generated
    random.seed(12)
    exp_demo = random.sample(demos, 5)
    exp_gen = random.sample(gens, 5)
    after = random.getstate()
    prm = RO.params_from_torch(ac.reward_net)
    p0 = ac._trainer.flat.double().cpu().numpy()
    f = lambda trajs, k: np.array([np.asarray(p[k], dtype=np.float32) for t in trajs for p in t], dtype=np.float64)
    (loss, first, second, _), g, _ = RO.irl_loss_and_grad(prm, f(exp_demo, 0), f(exp_demo, 1), f(exp_gen, 0), f(exp_gen, 1), 5, 5,
                                                         l1l2=True)
    gflat = RO.flatten_like_kernel(g)
    p_ref, _, _ = RO.adam_tf(p0, gflat, 0 * p0, 0 * p0, 1, lr=ac.lr_reward)
    random.seed(12)
    ac.update_reward()
    assert random.getstate() == after
    got = ac._trainer.flat.double().cpu().numpy()
    big = np.abs(gflat) > 1e-6 * np.abs(gflat).max()         # (Adam's first step is sign-like: see the several-steps test)
    assert np.max(np.abs(got - p_ref)[big]) <= 2e-2 * ac.lr_reward + 1e-6 * np.max(np.abs(p_ref))
    assert np.max(np.abs(got - p_ref)) <= 2.5 * ac.lr_reward
    assert abs(ac.loss_val - loss) <= 1e-5 * max(1, abs(loss)) and abs(ac.first_term_val - first) <= 1e-5
    assert abs(ac.second_term_val - second) <= 1e-5


def test_short_lists_use_everything_like_the_reference(dev):
    """Fewer than num_*_samples trajectories: the whole list is the batch (ac_irl.py:816-817, :826-827) and the first term
    still divides by num_demo_samples (:390)."""
    ac, demos = _irl(dev, reg='none', n_demo=3)
    ac.list_generated = ac.generate_trajectories(5)
    prm = RO.params_from_torch(ac.reward_net)
    f = lambda trajs, k: np.array([np.asarray(p[k], dtype=np.float32) for t in trajs for p in t], dtype=np.float64)
    st = random.getstate()
    ac.num_gen_samples = 5
    ac.update_reward()
    gens = ac.list_generated
    random.setstate(st)
    exp_gen = random.sample(gens, 5)
    (loss, first, second, _), _, _ = RO.irl_loss_and_grad(prm, f(demos, 0), f(demos, 1), f(exp_gen, 0), f(exp_gen, 1), 5, 5)
    assert abs(ac.loss_val - loss) <= 1e-5 * max(1, abs(loss)) and abs(ac.first_term_val - first) <= 1e-5


def test_outerloop_keeps_dsamp_on_the_device_and_list_view_is_lazy(dev):
    """outerloop (ac_irl.py:900-954) with the Philox sampler: D_samp is filled from mfg_rollout(WRITE_P) without a host copy,
    the FIFO keeps num_gen_from_policy * num_policies trajectories, and `list_generated` read afterwards is the reference's
    list[n] of list[15] of (pi [d], P [d,d]) with pi' = P^T pi along every trajectory."""
    ac, _ = _irl(dev, B=16)
    ac.outerloop(num_iterations=2, num_gen_from_policy=3, max_reward_iterations=4, max_forward_episodes=2, final_training=False)
    assert ac._gen_store._list is None                                   # nobody asked for the Python view
    lg = ac.list_generated
    assert len(lg) == 6 and len(lg[0]) == 15 and lg[0][0][0].shape == (15,) and lg[0][0][1].shape == (15, 15)
    assert lg[0][0][0].dtype == np.float64
    for traj in (lg[0], lg[5]):
        for t in range(14):
            assert np.allclose(traj[t + 1][0], traj[t][1].T.dot(traj[t][0]), rtol=1e-5, atol=1e-7)
    assert len(ac.list_eval_gen_transitions) == 90 and ac.list_eval_gen_transitions[16] is lg[1][1]
    assert ac.reward_update_count == 8 and ac._trainer.step_count == 8
    assert np.isfinite(ac.loss_val)


def test_eval_override_is_honoured(dev):
    """A caller-assigned evaluation list (e.g. get_eval_transitions, ac_irl.py:203-219) is what reward_iteration evaluates."""
    ac, demos = _irl(dev, reg='none')
    ac.list_generated = ac.generate_trajectories(6)
    ac.list_eval_gen_transitions = ac.get_eval_transitions(ac.list_generated)
    assert ac._eval_gen_override is not None and len(ac.list_eval_gen_transitions) == 6
    _, g_avg = ac._eval_reward_averages()
    s = np.array([np.asarray(p[0], dtype=np.float32) for p in ac.list_eval_gen_transitions], dtype=np.float64)
    a = np.array([np.asarray(p[1], dtype=np.float32) for p in ac.list_eval_gen_transitions], dtype=np.float64)
    ref = RO.forward(RO.params_from_torch(ac.reward_net), s, a).mean()
    assert abs(g_avg - ref) <= 1e-5
    ac.list_eval_gen_transitions = [p for t in ac.list_generated for p in t]
    assert ac._eval_gen_override is None


def test_in_place_growth_of_a_list_view_reaches_the_store(dev):
    """`ac.list_generated += more` / `.append(...)` bypass the property setter: the next update re-uploads a view whose length
    no longer matches its store (same for the demonstrations)."""
    ac, demos = _irl(dev, reg='none')
    ac.list_generated = ac.generate_trajectories(5)
    more = ac.generate_trajectories(2)
    ac.list_generated += more                              # getter + in-place extend + setter with the SAME object
    assert len(ac.list_generated) == 7
    ac.update_reward()
    assert len(ac._gen_store) == 7
    s, a = ac._gen_store.gather([6])
    assert np.array_equal(a[0, 3].cpu().numpy(), np.asarray(more[1][3][1], dtype=np.float32))
    ac.list_demonstrations.append(demos[0])
    ac.update_reward()
    assert len(ac._demo_store) == len(demos) == 8


def test_outerloop_with_the_reference_rng_goes_through_the_list_setters(dev):
    """rng='numpy' (the mode that retraces the reference's np.random consumption) generates trajectories on the host path and
    assigns lists: the FIFO `(list_generated + new)[k:]` of ac_irl.py:929-932 lands in the device store through the setter and
    the reward updates run on it."""
    rs = np.random.RandomState(4)
    d = 15
    mat = rs.dirichlet(np.ones(d), size=6)
    demos = [[(rs.dirichlet(np.ones(d)), rs.dirichlet(np.ones(d), size=d)) for _ in range(15)] for _ in range(6)]
    from discrete_mean_field_game_amd.ac_irl import AC_IRL
    np.random.seed(3); torch.manual_seed(3); random.seed(3)
    ac = AC_IRL(d=d, pi0=mat, demonstrations=demos, batch=1, rng='numpy', num_policies=2, seed=1, reg='none', verbose=0, device=dev)
    ac.outerloop(num_iterations=2, num_gen_from_policy=2, max_reward_iterations=3, max_forward_episodes=2, final_training=False)
    lg = ac.list_generated
    assert len(lg) == 4 and len(ac._gen_store) == 4 and ac._trainer.step_count == 6
    s, a = ac._gen_store.gather()
    assert np.array_equal(a[3, 14].cpu().numpy(), np.asarray(lg[3][14][1], dtype=np.float32))
    assert np.isfinite(ac.loss_val) and np.isfinite(float(np.ravel(ac.theta)[0]))


def test_reference_harnesses_test_convergence_and_test_reward_network(dev, tmp_path, monkeypatch):
    """ac_irl.py:961-1046: `test_convergence` (reward training against a fixed generated set, averages logged every iter_check
    updates, CSV in results/) and `test_reward_network` (average reward on train / test demonstrations and fresh trajectories)
    run on the device path; the logged averages equal an fp64 evaluation of the network at that point (no dropout here)."""
    monkeypatch.chdir(tmp_path)
    os.makedirs('results')
    ac, demos = _irl(dev, reg='l1l2')
    rows = ac.test_convergence(num_iterations=6, num_gen_from_policy=3, iter_check=3, filename='conv.csv')
    assert [r[0] for r in rows] == [3, 6] and ac._trainer.step_count == 6 and len(ac._gen_store) == 6
    text = open('results/conv.csv').read().splitlines()
    assert text[0] == 'iteration,reward_demo_avg,reward_gen_avg' and len(text) == 3 and text[2].startswith('6,')
    prm = RO.params_from_torch(ac.reward_net)
    f = lambda trajs, k: np.array([np.asarray(p[k], dtype=np.float32) for t in trajs for p in t], dtype=np.float64)
    ref_demo = RO.forward(prm, f(demos, 0), f(demos, 1)).mean()
    ref_gen = RO.forward(prm, f(ac.list_generated, 0), f(ac.list_generated, 1)).mean()
    assert abs(rows[-1][1] - ref_demo) <= 1e-5 and abs(rows[-1][2] - ref_gen) <= 1e-5
    ac.list_demonstrations_test = demos[:2]
    tr_avg, te_avg, ge_avg = ac.test_reward_network()
    assert abs(tr_avg - ref_demo) <= 1e-5 and len(ac._gen_store) == len(demos)
    assert abs(te_avg - RO.forward(prm, f(demos[:2], 0), f(demos[:2], 1)).mean()) <= 1e-5
    assert abs(ge_avg - RO.forward(prm, f(ac.list_generated, 0), f(ac.list_generated, 1)).mean()) <= 1e-5
    import inspect
    sig = inspect.signature(ac.evaluate)
    assert sig.parameters['d'].default == 15 and sig.parameters['outfile'].default == 'eval_mfg_round2/validation.csv'


def test_ragged_demonstrations_and_in_place_edits(dev):
    """(a) the reference's update_reward flattens demonstrations of ANY length (ac_irl.py:816-821): a list with other
    lengths than 15 pairs is accepted, stays on the host and trains through the autograd path; (b) replacing ONE trajectory of a
    list view in place (same length: `ac.list_demonstrations[i] = traj`) is noticed (object fingerprint) and re-uploaded."""
    from discrete_mean_field_game_amd.ac_irl import AC_IRL
    d = 15
    rs = np.random.RandomState(4)
    mat = rs.dirichlet(np.ones(d), size=6)
    traj = lambda n: [(rs.dirichlet(np.ones(d)), rs.dirichlet(np.ones(d), size=d)) for _ in range(n)]
    np.random.seed(11); torch.manual_seed(11); random.seed(3)
    ac = AC_IRL(d=d, pi0=mat, demonstrations=[traj(15), traj(9), traj(15), traj(20), traj(15), traj(15)], batch=8, seed=5, verbose=0)
    assert ac._demo_ragged and len(ac._demo_store) == 0 and len(ac.list_demonstrations) == 6
    ac.list_generated = [traj(15) for _ in range(5)]
    before = ac._trainer.flat.clone()
    steps0 = ac._trainer.step_count
    ac.update_reward()
    assert ac._trainer.step_count == steps0 + 1                      # ONE optimiser state: the autograd path stepped the trainer's Adam
    assert not torch.equal(before, ac._trainer.flat) and torch.isfinite(ac._trainer.flat).all()
    assert np.isfinite(ac.loss_val)
    rd, rg = ac._eval_reward_averages()                              # the whole ragged list is evaluated (15+9+15+20+15+15 pairs)
    assert np.isfinite(rd) and np.isfinite(rg)
    # (b) same-length in-place replacement of one demonstration
    ac2 = AC_IRL(d=d, pi0=mat, demonstrations=[traj(15) for _ in range(6)], batch=8, seed=5, verbose=0)
    assert not ac2._demo_ragged and len(ac2._demo_store) == 6
    new = traj(15)
    ac2.list_demonstrations[2] = new
    ac2._resync_stores()
    s, a = ac2._demo_store.gather([2])
    assert np.allclose(s[0, 0].cpu().numpy(), new[0][0].astype(np.float32)) and np.allclose(a[0, 14].cpu().numpy(), new[14][1].astype(np.float32))


def test_checkpoint_without_trainer_state_takes_the_torch_adam_moments(dev):
    """a checkpoint written before the HIP training step existed carries its Adam state in the torch optimiser only;
    loading it seeds the trainer's moments and step count from there instead of silently starting from zero."""
    from discrete_mean_field_game_amd.ac_irl import AC_IRL
    d = 15
    rs = np.random.RandomState(4)
    mat = rs.dirichlet(np.ones(d), size=6)
    demos = [[(rs.dirichlet(np.ones(d)), rs.dirichlet(np.ones(d), size=d)) for _ in range(15)] for _ in range(6)]
    np.random.seed(11); torch.manual_seed(11)
    ac = AC_IRL(d=d, pi0=mat, demonstrations=demos, batch=8, seed=5, verbose=0, reg='none')
    # an "old" run: three torch-Adam steps on a dummy loss give the optimiser a state
    for _ in range(3):
        ac.optimizer.zero_grad()
        loss = sum((p ** 2).sum() for p in ac.reward_net.parameters())
        loss.backward()
        ac.optimizer.step()
    st = ac.state_dict()
    st.pop('reward_trainer', None)
    ac2 = AC_IRL(d=d, pi0=mat, demonstrations=demos, batch=8, seed=5, verbose=0, reg='none')
    ac2.load_state_dict(st)
    assert ac2._trainer.step_count == 3
    off = 0
    for p in ac.reward_net.parameters():
        n = p.numel()
        assert torch.allclose(ac2._trainer.m[off:off + n], ac.optimizer.state[p]['exp_avg'].reshape(-1))
        assert torch.allclose(ac2._trainer.v[off:off + n], ac.optimizer.state[p]['exp_avg_sq'].reshape(-1))
        off += n
