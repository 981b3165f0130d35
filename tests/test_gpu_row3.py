"""-m gpu: the two lane mappings of the d = 21 sampling launches give the SAME BITS.

Round 6: batches that under-fill the machine run `k_core_row3` -- one trajectory per wavefront, three lanes
per matrix row (csrc/mfg_core_row3.hip) -- instead of the packed `k_core_small` (three trajectories per wavefront, a lane per
row; hot loop of the reference's mfg_ac2.py:478-526).  Which kernel a launch takes is a function of the batch a rank holds, so
it must never show in the results: the same Philox quads keyed by the same element ids, ONE summation tree for the row sums,
the column pass, the value and the per-trajectory sums.  `mfg_set_core_mapping` (include/mfg_hip.h) forces either mapping;
every output is compared with array_equal.  The oracle parity of the new kernel follows from the packed kernel's
(tests/test_gpu_parity.py, test_gpu_fullsize.py run it wherever the batch is small) and is re-checked here directly.
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU')
    from discrete_mean_field_game_amd import ops
    ops.init()
    return torch.device('cuda', 0)


@pytest.fixture(autouse=True)
def _auto_mapping():
    yield
    from discrete_mean_field_game_amd import _lib as L
    L.lib().mfg_set_core_mapping(0)


def _O():
    from oracle import mfg_oracle
    return mfg_oracle


def _both(fn):
    """fn() under the packed mapping (1) and the one-trajectory-per-wave mapping (2)."""
    from discrete_mean_field_game_amd import _lib as L
    out = []
    for mode in (1, 2):
        L.lib().mfg_set_core_mapping(mode)
        out.append(fn())
    L.lib().mfg_set_core_mapping(0)
    return out


def _same(a, b, keys):
    for k in keys:
        x, y = a[k], b[k]
        if x is None and y is None:
            continue
        assert torch.equal(x, y), 'output %r differs between the lane mappings (max |diff| %g)' % (
            k, float((x.double() - y.double()).abs().max()))


@pytest.mark.parametrize('B,T,first_step', [(1, 15, 0), (2, 3, 1), (3, 2, 4), (4, 15, 7), (5, 1, 0), (5, 1, 3), (64, 15, 0),
                                            (257, 6, 11), (1000, 15, 2), (4096, 15, 5)])
@pytest.mark.parametrize('td,write_P,discount_pow,reward_kind', [(True, False, False, 0), (True, True, True, 0), (False, True, False, 0),
                                                                  (False, False, False, 0), (True, False, False, 1)])
@pytest.mark.parametrize('d', [21, 15])
def test_rollout_is_bit_identical_in_both_mappings(dev, d, B, T, first_step, td, write_P, discount_pow, reward_kind):
    from discrete_mean_field_game_amd import ops
    rs = np.random.RandomState(1000 + B + T)
    pi0 = torch.as_tensor(rs.dirichlet(np.ones(d), size=B).astype(np.float32), device=dev)
    th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
    w = torch.as_tensor(rs.rand(ops.num_features(d)), device=dev) if td else None

    def run():
        return ops.rollout(pi0, T, th, 0.16, 12000.0, w=w, gamma=0.9, reward_kind=reward_kind, seed=99, first_step=first_step,
                           traj_offset=12345678901, td=td, write_P=write_P, discount_pow=discount_pow)
    a, b = _both(run)
    _same(a, b, ['pi_traj', 'pi_last', 'reward', 'delta', 'g', 'P', 'G'])
    assert torch.isfinite(a['pi_traj']).all()


def test_row3_kernel_vs_the_oracle_directly(dev):
    """The new kernel against the fp64 oracle on its own sampled actions (the packed kernel's parity test, with the mapping forced)."""
    from discrete_mean_field_game_amd import ops, _lib as L
    O = _O()
    d, B, T = 21, 100, 15
    rs = np.random.RandomState(31 + d + B)
    theta, shift, scale, gamma = 8.86349, 0.16, 12000.0, 0.9
    pi0 = rs.dirichlet(np.ones(d), size=B).astype(np.float32)
    w = rs.rand(O.num_features(d))
    L.lib().mfg_set_core_mapping(2)
    out = ops.rollout(torch.as_tensor(pi0, device=dev), T, torch.tensor([theta], dtype=torch.float64, device=dev), shift, scale,
                      w=torch.as_tensor(w, device=dev), gamma=gamma, seed=99, first_step=5, traj_offset=1000, td=True, write_P=True)
    P = out['P'].cpu().numpy()
    assert np.max(np.abs(P.astype(np.float64).sum(-1) - 1)) < 5e-7
    traj = O.batched_rollout_given_P(pi0, P, w, theta, shift, gamma=gamma)[0]
    assert np.allclose(out['pi_traj'].cpu().numpy(), traj, rtol=3e-7, atol=1e-12)
    pt = out['pi_traj'].cpu().numpy().astype(np.float64)
    r_ref = np.stack([O.calc_reward(P[:, t].astype(np.float64), pt[:, t]) for t in range(T)], 1)
    assert np.max(np.abs(out['reward'].cpu().numpy() - r_ref) / np.maximum(np.abs(r_ref), 1e-30)) < 1e-6      # bar: 1e-5 relative
    g_ref = np.stack([O.calc_gradient(P[:, t], pt[:, t], theta, shift) for t in range(T)], 1)
    assert np.max(np.abs(out['g'].cpu().numpy() - g_ref) / np.maximum(np.abs(g_ref), 1e-30)) < 1e-5           # mixed precision
    V = O.calc_features(pt).dot(w)
    d_ref = r_ref + gamma * V[:, 1:] - V[:, :-1]
    assert np.max(np.abs(out['delta'].cpu().numpy() - d_ref)) < 1e-11 * max(1.0, np.abs(V).max())
    # the given-P kernel on the materialised actions reproduces pi' and the reward bit for bit (one summation tree at d = 21)
    pn, r1 = ops.step_given_P(out['pi_traj'][:, 0].contiguous(), out['P'][:, 0].contiguous())
    assert torch.equal(pn, out['pi_traj'][:, 1]) and torch.equal(r1, out['reward'][:, 0])
    # ... and the stand-alone sampler draws the same actions in either mapping
    for mode in (1, 2):
        L.lib().mfg_set_core_mapping(mode)
        P0 = ops.sample_dirichlet(torch.as_tensor(pi0, device=dev), torch.tensor([theta], dtype=torch.float64, device=dev), shift, scale,
                                  seed=99, step=5, traj_offset=1000)
        assert np.array_equal(P0.cpu().numpy(), P[:, 0])


@pytest.mark.parametrize('d', [21, 15])
def test_small_shapes_and_cold_paths_in_both_mappings(dev, d):
    """Policies whose concentrations fall below 1 (the U^(1/a) boost) and whose acceptance tests go to the exact path: the
    continuation draws are keyed by the ELEMENT, so the trailing element that runs through the quad code must find its own."""
    from discrete_mean_field_game_amd import ops
    B, T = 300, 4
    rs = np.random.RandomState(5)
    pi0 = torch.as_tensor(rs.dirichlet(0.3 * np.ones(d), size=B).astype(np.float32), device=dev)
    for theta, scale in ((8.86349, 0.7), (2.0, 3.0), (20.0, 40.0), (8.86349, 1.0e5)):
        th = torch.tensor([theta], dtype=torch.float64, device=dev)
        w = torch.as_tensor(rs.rand(ops.num_features(d)), device=dev)
        for fs in (0, 1):
            a, b = _both(lambda: ops.rollout(pi0, T, th, 0.16, scale, w=w, gamma=1.0, seed=3, first_step=fs, td=True, write_P=True))
            _same(a, b, ['pi_traj', 'reward', 'delta', 'g', 'P'])
            assert torch.isfinite(a['P']).all() and float((a['P'].double().sum(-1) - 1).abs().max()) < 5e-7


@pytest.mark.parametrize('mode', ['rollout', 'step'])
@pytest.mark.parametrize('d', [21, 15])
def test_class_training_does_not_depend_on_the_mapping(dev, d, mode):
    """actor_critic.train over a batch the automatic choice gives to the new kernel: parameters, returns and final states equal
    the run with the packed kernel forced, bit for bit (native episode loops, in-kernel start draw, deferred nothing)."""
    from discrete_mean_field_game_amd import _lib as L
    from discrete_mean_field_game_amd.mfg_ac2 import actor_critic
    rs = np.random.RandomState(0)
    mat = rs.dirichlet(np.ones(d), size=16)
    res = []
    for m in (1, 0):
        L.lib().mfg_set_core_mapping(m)
        np.random.seed(7)
        ac = actor_critic(d=d, pi0=mat, batch=700, rng='philox', seed=5, update_every=mode, verbose=0)
        ac.train(num_episodes=4, gamma=0.9)
        res.append((float(np.ravel(ac.theta)[0]), ac.w[:, 0].copy(), ac._last_pi.cpu().numpy().copy()))
    L.lib().mfg_set_core_mapping(0)
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    assert res[0][0] != 8.86349


@pytest.mark.parametrize('d', [21, 15])
def test_irl_training_does_not_depend_on_the_mapping(dev, d):
    """AC_IRL.train in rollout mode (mfg_train_rollout_irl: rollout with the actions written out | reward network | sums)."""
    from discrete_mean_field_game_amd import _lib as L
    from discrete_mean_field_game_amd.ac_irl import AC_IRL
    rs = np.random.RandomState(0)
    mat = rs.dirichlet(np.ones(d), size=16)
    res = []
    for m in (1, 2):
        L.lib().mfg_set_core_mapping(m)
        np.random.seed(5); torch.manual_seed(5)
        ac = AC_IRL(theta=8.64, shift=0.0, alpha_scale=1e4, d=d, pi0=mat, demonstrations=[], batch=300, seed=3,
                    update_every='rollout', verbose=0)
        ac.train(max_episodes=3, stop_criteria=-1)
        res.append((float(np.ravel(ac.theta)[0]), ac.w[:, 0].copy()))
    L.lib().mfg_set_core_mapping(0)
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and res[0][0] != 8.64


@pytest.mark.parametrize('B,T_odd', [(300, True), (4096, True), (50, False)])
@pytest.mark.parametrize('d', [21, 15])
def test_irl_env_step_variants_do_not_depend_on_the_mapping(dev, d, B, T_odd):
    """AC_IRL.train in step mode (the class default): mfg_train_episode_irl_draw = per env step [STEP variant of the sampling kernel:
    theta from the previous step's partial rows, their reduction in the grid's last blocks | reward network + TD error + batch
    sums].  The STEP variants exist in both lane mappings; parameters, returns and final states must not depend on which one runs.
    (drawn start states: STEP = 2 at the first step, STEP = 1 afterwards; the native call itself with a given start state too)"""
    from discrete_mean_field_game_amd import ops, _lib as L
    from discrete_mean_field_game_amd.ac_irl import AC_IRL
    rs = np.random.RandomState(0)
    mat = rs.dirichlet(np.ones(d), size=16)
    res = []
    for m in (1, 2):
        L.lib().mfg_set_core_mapping(m)
        np.random.seed(5); torch.manual_seed(5)
        ac = AC_IRL(theta=8.64, shift=0.0, alpha_scale=1e4, d=d, pi0=mat, demonstrations=[], batch=B, seed=3, update_every='step',
                    verbose=0)
        ac.train(max_episodes=2, stop_criteria=-1, gamma=0.95)
        res.append((float(np.ravel(ac.theta)[0]), ac.w[:, 0].copy(), ac._last_pi.cpu().numpy().copy() if hasattr(ac, '_last_pi') and ac._last_pi is not None else None))
    L.lib().mfg_set_core_mapping(0)
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and res[0][0] != 8.64
    if res[0][2] is not None:
        assert np.array_equal(res[0][2], res[1][2])


@pytest.mark.parametrize('d', [21, 15])
def test_deferred_update_chain_in_both_mappings(dev, d):
    """mfg_train_rollout_deferred (the multi-rank cycle: the previous update applied while the weights are staged, block 0
    publishing the new parameters): three chained episodes, identical parameters / sums / outputs in either mapping."""
    from discrete_mean_field_game_amd import ops
    B, T = 500, 15
    rs = np.random.RandomState(2)
    mat = torch.as_tensor(rs.dirichlet(np.ones(d), size=9).astype(np.float32), device=dev)
    F = ops.num_features(d)

    def run():
        theta = torch.tensor([8.86349], dtype=torch.float64, device=dev)
        w = torch.as_tensor(np.random.RandomState(1).rand(F), device=dev)
        ta, wa = torch.empty_like(theta), torch.empty_like(w)
        G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
        ws = ops.workspace(B * T, d, dev)
        bufs = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'pi_last': torch.empty(B, d, device=dev),
                'reward': torch.empty(B, T, device=dev), 'delta': torch.empty(B, T, dtype=torch.float64, device=dev),
                'g': torch.empty(B, T, dtype=torch.float64, device=dev)}
        racc = torch.zeros(4, dtype=torch.float64, device=dev)
        pending = None
        snaps = []
        for ep in range(3):
            ops.train_rollout_deferred(mat, None, T, theta, w, pending, ta, wa, 0.16, 12000.0, 0.9, G, ws, bufs, seed=8,
                                       first_step=ep * T, traj_offset=77)
            if pending is not None:
                theta, ta = ta, theta
                w, wa = wa, w
            pending = (G, 0.1 / (ep + 1), 0.001 / (ep + 1), racc.data_ptr() + 8 * ep)
            snaps.append((theta.clone(), w.clone(), G.clone(), bufs['delta'].clone(), bufs['pi_last'].clone()))
        return {'theta': torch.cat([s_[0] for s_ in snaps]), 'w': torch.cat([s_[1] for s_ in snaps]),
                'G': torch.cat([s_[2] for s_ in snaps]), 'delta': torch.cat([s_[3] for s_ in snaps]),
                'pi_last': torch.cat([s_[4] for s_ in snaps]), 'racc': racc.clone()}
    a, b = _both(run)
    _same(a, b, ['theta', 'w', 'G', 'delta', 'pi_last', 'racc'])
    assert not torch.equal(a['theta'][0:1], a['theta'][2:3])


def test_forced_new_mapping_beyond_one_resident_grid(dev):
    """Forced onto a batch larger than the grid cap (8 x the resident blocks: 32 768 trajectories on 256 CUs) a wave of the new
    kernel walks several trajectories (the prefetch of the next start state, the per-trajectory re-initialisation): same bits."""
    from discrete_mean_field_game_amd import ops
    d, B, T = 21, 70001, 2
    rs = np.random.RandomState(9)
    mat = torch.as_tensor(rs.dirichlet(np.ones(d), size=50).astype(np.float32), device=dev)
    pi0 = mat[torch.as_tensor(rs.randint(50, size=B), device=dev)].contiguous()
    th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
    w = torch.as_tensor(rs.rand(ops.num_features(d)), device=dev)
    a, b = _both(lambda: ops.rollout(pi0, T, th, 0.16, 12000.0, w=w, gamma=0.9, seed=4, first_step=3, td=True))
    _same(a, b, ['pi_traj', 'pi_last', 'reward', 'delta', 'g', 'G'])


def test_automatic_choice_follows_the_batch_size(dev):
    """mode 0: batches up to one resident round of the new kernel (16 trajectories per CU) take it, larger ones the packed kernel --
    observable only through timing, so the check here is that results across the threshold stay those of the forced modes."""
    from discrete_mean_field_game_amd import ops, _lib as L
    import ctypes as C
    cus = C.c_int(0)
    L.check(L.lib().mfg_device_info(C.byref(cus), None, 0), 'mfg_device_info')
    d, T = 21, 2
    th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
    for B in (16 * cus.value, 16 * cus.value + 1):
        pi0 = torch.full((B, d), 1.0 / d, device=dev)
        L.lib().mfg_set_core_mapping(0)
        auto = ops.rollout(pi0, T, th, 0.16, 12000.0, seed=1, td=False)
        a, b = _both(lambda: ops.rollout(pi0, T, th, 0.16, 12000.0, seed=1, td=False))
        assert torch.equal(auto['pi_traj'], a['pi_traj']) and torch.equal(a['pi_traj'], b['pi_traj'])
