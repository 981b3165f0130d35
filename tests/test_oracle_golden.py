"""Pin the CPU oracle against vectors produced by the unmodified reference.

Fixtures: tests/golden/*.npz written by oracle/gen_golden.py (which imports
/root/reference in the build container).  Nothing here reads /root/reference.
"""
import glob
import os

import numpy as np
import pytest

from oracle import mfg_oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


# ---- the reference's own known-answer inputs (test2.py) ---------------------------
def test_kat_reward():
    k = load('kat_mfg_ac2.npz')
    r = O.calc_reward(k['reward_P'], k['reward_pi'])
    assert r.shape == (1,)                                  # reference returns shape (1,)
    assert np.array_equal(r, k['reward_out'])
    assert abs(r[0] - (-39.07)) < 1e-12


def test_kat_value_and_features():
    k = load('kat_mfg_ac2.npz')
    f = O.calc_features(k['value_pi'])
    assert np.array_equal(f, k['value_features'])
    assert np.allclose(f, [.01, .02, .07, .04, .14, .49, .1, .2, .7, 1.], rtol=0, atol=1e-15)
    v = O.calc_value(k['value_pi'], np.ones(10))
    assert v == k['value_out']
    assert abs(v - 2.77) < 1e-12


def test_kat_jsd():
    k = load('kat_mfg_ac2.npz')
    p, q = k['jsd_p'].copy(), k['jsd_q'].copy()
    j = O.JSD(p, q)
    assert j == k['jsd_out']
    assert abs(j - 0.34858446189521375) < 1e-15
    assert p[2] == 0.0                                       # oracle does not mutate its input
    # batched form
    jb = O.JSD(np.stack([p, q]), np.stack([q, p]))
    assert np.allclose(jb, j, rtol=1e-15)


def test_kat_gradient_three_way():
    k = load('kat_mfg_ac2.npz')
    pi, P = k['grad_pi'], k['grad_P']
    assert np.array_equal(O.calc_alpha(pi, 10, 0.4), k['grad_alpha'])
    assert np.array_equal(O.calc_alpha_deriv(pi, 10, 0.4), k['grad_alpha_deriv'])
    g = O.calc_gradient(P, pi, 10, 0.4)
    assert g == k['grad_vectorized']
    assert abs(g - k['grad_basic']) < 1e-12 and abs(g - k['grad_loop']) < 1e-12
    assert abs(g - (-6.302201890992953)) < 1e-12
    assert np.array_equal(O.calc_reward(P, pi), k['grad_reward'])
    assert np.array_equal(O.transition(P, pi), k['grad_pi_next'])
    assert np.allclose(P.sum(1), 1, atol=1e-14) and abs(O.transition(P, pi).sum() - 1) < 1e-14
    # seeded sampler reproduces the reference's P bit for bit
    np.random.seed(42)
    assert np.array_equal(O.sample_action(pi, 10, 0.4, 12000), P)


# ---- per-function vectors ------------------------------------------------------------
@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(G, 'functions_cfg*.npz'))))
def test_functions(path):
    z = np.load(path)
    d, theta, shift, scale = int(z['d']), float(z['theta']), float(z['shift']), float(z['alpha_scale'])
    w = z['w']
    for k in range(z['pi'].shape[0]):
        pi, P = z['pi'][k], z['P'][k]
        assert np.array_equal(O.calc_alpha(pi, theta, shift), z['alpha'][k])
        assert np.array_equal(O.calc_alpha_deriv(pi, theta, shift), z['alpha_deriv'][k])
        np.random.seed(int(z['gamma_seed'][k]))
        assert np.array_equal(O.sample_action(pi, theta, shift, scale), P)
        assert np.array_equal(O.transition(P, pi), z['pi_next'][k])
        assert O.calc_reward(P, pi)[0] == z['reward'][k]
        assert np.array_equal(O.calc_features(pi), z['features'][k])
        assert O.calc_value(pi, w)[0] == z['value'][k]
        assert O.calc_value(z['pi_next'][k], w)[0] == z['value_next'][k]
        g = O.calc_gradient(P, pi, theta, shift)
        assert g == z['gradient'][k]
    # batched forms agree with the per-sample reference values
    pi, P = z['pi'], z['P']
    assert np.allclose(O.transition(P, pi), z['pi_next'], rtol=1e-14, atol=1e-300)
    assert np.allclose(O.calc_reward(P, pi), z['reward'], rtol=1e-9, atol=1e-18)
    assert np.allclose(O.calc_gradient(P, pi, theta, shift), z['gradient'], rtol=1e-12)
    delta, g, G_w, G_theta, _ = O.batched_td_pg(pi, z['pi_next'], P, z['reward'], w, theta, shift, 1.0)
    assert np.allclose(delta, z['delta'], rtol=1e-12, atol=1e-15)
    assert np.allclose(G_w, (z['delta'][:, None] * z['features']).sum(0), rtol=1e-12, atol=1e-15)
    assert np.isclose(G_theta, np.sum(z['delta'] * z['gradient']), rtol=1e-12)


def test_reward_synthetic_variant():
    z = load('reward_mfg_synthetic.npz')
    assert np.allclose(O.calc_reward_synthetic(z['P'], z['pi']), z['reward'], rtol=1e-14)
    kat = O.calc_reward_synthetic(np.array([[1, 3, 3], [4, 5, 6], [7, 8, 9.]]), np.array([.1, .2, .7]))
    assert abs(kat - float(z['kat_reward'])) < 1e-12 and abs(kat - (-76.55)) < 1e-12


# ---- integer bookkeeping: bit exact -----------------------------------------------------
def test_int_bookkeeping():
    z = load('ints_mfg_ac2.npz')
    out, order = O.reorder([list(r) for r in z['reorder_in']])
    assert np.array_equal(np.array(out), z['reorder_out'])
    assert order == [1, 3, 6, 4, 0, 2, 5]                   # stable on the tie 9,9
    for d in (3, 4, 21):
        pairs = z['pairs_d%d' % d]
        tab = O.feature_index_table(d)
        for k, (i, j) in enumerate(pairs):
            assert O.feature_index(int(i), int(j), d) == k
            assert tab[i, j] == k and tab[j, i] == k
        assert O.num_features(d) == len(pairs) + d + 1
    np.random.seed(7)
    seq = [np.random.randint(z['mat_pi0_d21'].shape[0]) for _ in range(32)]
    assert np.array_equal(np.array(seq), z['randint_seq'])
    line = ' '.join('%.3e' % v for v in z['mat_pi0_d21'][0]) + ' 1.000e-03\n'
    assert np.array_equal(np.array(O.parse_pi0_line(line, 21)), z['mat_pi0_d21'][0])


# ---- seeded train() traces ----------------------------------------------------------------
@pytest.mark.parametrize('name', ['c0_g1', 'c1_g1', 'c0_g09', 'c1_g09'])
def test_train_trace_mfg_ac2(name):
    z = load('train_mfg_ac2_%s.npz' % name)
    np.random.seed(int(z['seed']))
    w0 = np.random.rand(O.num_features(21), 1)               # the constructor's init_w draw
    assert np.array_equal(w0, z['w0'])
    out = O.train_mfg_ac2(z['mat_pi0'], w0, float(z['theta0']), float(z['shift']), float(z['alpha_scale']),
                          int(z['num_episodes']), gamma=float(z['gamma']), constant=int(z['constant']),
                          lr_critic=float(z['lr_critic']), lr_actor=float(z['lr_actor']))
    assert np.array_equal(out['P'], z['P'])                  # same RNG consumption order
    assert np.array_equal(out['pi'], z['pi'])
    theta_before = np.concatenate([[float(z['theta0'])], out['theta'][:-1]])
    assert np.array_equal(theta_before, z['theta_before'])   # bit-for-bit theta after every step
    assert out['theta_final'] == float(z['theta_final'])
    assert np.array_equal(out['w_final'], z['w_final'])


@pytest.mark.parametrize('name', ['c0_g1', 'c1_g09', 'c0_g09_stop'])
def test_train_trace_ac_irl(name):
    from oracle.gen_golden import fake_reward
    z = load('train_ac_irl_%s.npz' % name)
    np.random.seed(int(z['seed']))
    w0 = np.random.rand(O.num_features(21), 1)
    assert np.array_equal(w0, z['w0'])
    out = O.train_ac_irl(z['mat_pi0'], w0, float(z['theta0']), float(z['shift']), float(z['alpha_scale']),
                         int(z['max_episodes']), lambda pi, P: np.array([[fake_reward(pi, P)]]),
                         stop_criteria=float(z['stop_criteria']), gamma=float(z['gamma']),
                         constant=bool(z['constant']), lr_critic=float(z['lr_critic']),
                         lr_actor=float(z['lr_actor']))
    n = int(z['steps_run'])
    assert len(out['theta']) == n                            # same early-stop episode
    theta_before = np.concatenate([[float(z['theta0'])], out['theta'][:-1]])
    assert np.array_equal(theta_before, z['theta_before'])
    assert out['theta_final'] == float(z['theta_final'])
    assert np.array_equal(out['w_final'], z['w_final'])


def test_generate_trajectory():
    z = load('generate_trajectory_mfg_ac2.npz')
    np.random.seed(int(z['seed']))
    t = O.generate_trajectory(z['pi0'], int(z['total_hours']), float(z['theta']), float(z['shift']),
                              float(z['alpha_scale']))
    assert np.array_equal(t, z['traj'])
    assert np.allclose(t.sum(1), t[0].sum(), atol=1e-12)      # mass is conserved, not renormalised


def test_generate_trajectories_ac_irl():
    z = load('generate_trajectories_ac_irl.npz')
    np.random.seed(int(z['seed']))
    trajs = O.generate_trajectories(int(z['n']), z['mat_pi0'], float(z['theta']), float(z['shift']),
                                    float(z['alpha_scale']))
    for b, traj in enumerate(trajs):
        assert len(traj) == 15
        for t, (pi, P) in enumerate(traj):
            assert np.array_equal(pi, z['pi'][b, t]) and np.array_equal(P, z['P'][b, t])
    g = O.calc_gradient(z['P'], z['pi'], float(z['theta']), float(z['shift']))
    assert np.allclose(g, z['gradient'], rtol=1e-12)


def test_batched_rollout_matches_sequential_semantics():
    """B=1 'rollout' sums equal the per-step increments of the reference loop at fixed (theta, w)."""
    z = load('train_mfg_ac2_c1_g1.npz')
    P = z['P'][:15][None]                                     # first episode, [1,15,d,d]
    pi0 = z['pi'][0][None]
    w = z['w0'][:, 0]
    traj, R, D, Gs, G_w, G_theta = O.batched_rollout_given_P(pi0, P, w, float(z['theta0']), float(z['shift']),
                                                             storage_dtype=np.float64)
    assert np.allclose(traj[0, :15], z['pi'][:15], rtol=1e-12, atol=1e-300)
    phi = O.calc_features(traj[0, :15])
    assert np.allclose(G_w, (D[0][:, None] * phi).sum(0), rtol=1e-12)
    assert np.isclose(G_theta, float(np.sum(D[0] * Gs[0])), rtol=1e-12)


def test_backward_value_recursion_mfg_synthetic():
    """evaluate_synthetic / evaluate_synthetic_JSD of the reference (mfg_synthetic.py:741-899) on captured actions."""
    z = load('backward_value_mfg_synthetic.npz')
    assert np.allclose(O.calc_reward_vector(z['actions_l1'][0, 0]), z['reward_vector'], rtol=1e-14)
    V, l1, _ = O.evaluate_synthetic_diffs(z['actions_l1'])
    assert V.shape == (4, 16, 21) and np.all(V[:, 15] == 0)
    assert np.isclose(l1.mean(), float(z['l1_mean']), rtol=1e-12) and np.isclose(l1.std(), float(z['l1_std']), rtol=1e-12)
    _, _, jsd = O.evaluate_synthetic_diffs(z['actions_jsd'])
    assert np.isclose(jsd.mean(), float(z['jsd_mean']), rtol=1e-12) and np.isclose(jsd.std(), float(z['jsd_std']), rtol=1e-12)


def test_policy_logpdf_against_scipy_dirichlet():
    """f1 (unpinned TF code, ac_irl.py:270-289): the oracle's log-density equals the sum of scipy's Dirichlet row
    log-pdfs with the reference's alpha matrix, and calc_z's log-space result equals the direct formula (:292-321)."""
    from scipy.stats import dirichlet
    from scipy.special import logsumexp
    rs = np.random.RandomState(0)
    d, N = 5, 6
    pi = rs.dirichlet(np.ones(d), size=N)
    P = rs.dirichlet(np.ones(d) * 2.0, size=(N, d))
    thetas = [3.0, 6.5]
    lq = O.policy_logpdf(pi, P, thetas, 0.1)
    for n in range(N):
        for k, th in enumerate(thetas):
            mat1 = np.repeat(pi[n].reshape(1, d), d, 0)
            mat2 = np.repeat(pi[n].reshape(d, 1), d, 1)
            alpha = np.log(1 + np.exp(th * (mat1 - mat2 - 0.1)))
            want = sum(dirichlet.logpdf(P[n, i], alpha[i]) for i in range(d))
            assert abs(lq[n, k] - want) < 1e-9 * max(1.0, abs(want))
    # calc_z: 2 trajectories x 3 steps, direct product formula in linear space (small enough not to overflow)
    M, T = 2, 3
    lz = O.calc_z(pi.reshape(M, T, d), P.reshape(M, T, d, d), thetas, 0.1, num_start_samples=7)
    lqf = O.policy_logpdf(pi, P, thetas, 0.1, 1.0, 1.0 + 1e-6).reshape(M, T, 2)
    q = np.exp(lqf).prod(1) / 7.0
    z = len(thetas) / q.sum(1)
    np.testing.assert_allclose(np.exp(lz), z, rtol=1e-10)
