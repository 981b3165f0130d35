"""-m gpu parity tests: the HIP path (through the C ABI) against the CPU oracle and the reference's
golden vectors.  Tolerances (BASELINE.json north star): bit exact for integer / index work, <= 1e-5
relative for fp32 rewards / returns; tighter where the kernel's fp64 accumulation allows.
"""
import glob
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU: the HIP path has no CPU fallback')
    return torch.device('cuda:0')


def ops():
    from discrete_mean_field_game_amd import ops as _ops
    return _ops


def O():
    from oracle import mfg_oracle
    return mfg_oracle


def t32(x, dev):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32), device=dev)


def t64(x, dev):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=dev)


def rel(a, b, floor=0.0):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)) if a.size else 0.0


def rand_case(rs, B, d, conc=1.0, pconc=1.0):
    pi = rs.dirichlet(np.ones(d) * conc, size=B).astype(np.float32)
    P = rs.dirichlet(np.ones(d) * pconc, size=(B, d)).astype(np.float32)
    return pi, P


# ---------------------------------------------------------------------------------------------------
def test_library_loaded_and_arch(dev):
    from discrete_mean_field_game_amd import _lib as L
    import ctypes as C
    cu = C.c_int(0)
    buf = C.create_string_buffer(64)
    L.check(L.lib().mfg_device_info(C.byref(cu), buf, 64), 'mfg_device_info')
    assert cu.value > 0 and buf.value.decode().startswith('gfx')


def test_philox_bit_exact(dev):
    from oracle.philox_ref import philox4x32_10
    seed = 0x9E3779B97F4A7C15
    out = ops().philox_raw(seed, 12345, 7, 0xDEADBEEF, 0x10002, 4096, dev).cpu().numpy().view(np.uint32)
    ref = philox4x32_10(np.arange(12345, 12345 + 4096), 7, 0xDEADBEEF, 0x10002, seed & 0xFFFFFFFF, seed >> 32)
    assert np.array_equal(out, np.stack(ref, axis=1))


@pytest.mark.parametrize('d,B', [(3, 1), (3, 257), (4, 1000), (21, 1), (21, 13), (21, 4096), (32, 100), (47, 33),
                                 (64, 50), (65, 7), (100, 9), (128, 64), (130, 5), (192, 6), (256, 32), (320, 3),
                                 (512, 2)])
@pytest.mark.parametrize('kind', [0, 1])
def test_step_given_P(dev, d, B, kind):
    rs = np.random.RandomState(d * 1000 + B)
    pi, P = rand_case(rs, B, d)
    pn, r = ops().step_given_P(t32(pi, dev), t32(P, dev), reward_kind=kind)
    ref_pn = O().transition(P, pi)
    ref_r = O().calc_reward(P, pi) if kind == 0 else O().calc_reward_synthetic(P, pi)
    assert np.array_equal(pn.cpu().numpy(), ref_pn.astype(np.float32))   # fp64 accumulate, one rounding
    # fp32 reward: 1e-5 relative (north star); fp64 accumulation gives ~6e-8 + cancellation-free
    assert rel(r.cpu().numpy(), ref_r, floor=1e-30) < 1e-6


def test_step_given_P_transition_only(dev):
    rs = np.random.RandomState(3)
    pi, P = rand_case(rs, 100, 21)
    pn, r = ops().step_given_P(t32(pi, dev), t32(P, dev), want_reward=False)
    assert r is None
    assert np.array_equal(pn.cpu().numpy(), O().transition(P, pi).astype(np.float32))


def test_reference_kats_on_device(dev):
    k = np.load(os.path.join(G, 'kat_mfg_ac2.npz'))
    # test2.py:46-56 reward KAT (P is not stochastic there: exercises the raw formula)
    pn, r = ops().step_given_P(t32(k['reward_pi'][None], dev), t32(k['reward_P'][None], dev))
    assert abs(float(r[0]) - (-39.07)) < 1e-5 * 39.07
    # test2.py:73-88 value KAT
    v = ops().value(t32(k['value_pi'][None], dev), t64(np.ones(10), dev))
    assert abs(float(v[0]) - 2.77) < 1e-6
    f = ops().features(t32(k['value_pi'][None], dev)).cpu().numpy()[0]
    assert np.allclose(f, k['value_features'], rtol=1e-6)
    # JSD KAT
    j = ops().jsd(t32(k['jsd_p'][None], dev), t32(k['jsd_q'][None], dev))
    assert abs(float(j[0]) - 0.34858446189521375) < 1e-6
    # test2.py:105-121 gradient three-way KAT (seed 42 P captured from the reference)
    theta = t64([10.0], dev)
    g = ops().score(t32(k['grad_pi'][None], dev), t32(k['grad_P'][None], dev), theta, 0.4, precision='f64')
    g_ref32 = O().calc_gradient(k['grad_P'].astype(np.float32), k['grad_pi'].astype(np.float32), 10.0, 0.4)
    assert abs(float(g[0]) - g_ref32) < 1e-9 * abs(g_ref32)
    gm = ops().score(t32(k['grad_pi'][None], dev), t32(k['grad_P'][None], dev), theta, 0.4, precision='mixed')
    assert abs(float(gm[0]) - g_ref32) < 1e-5 * abs(g_ref32)
    assert abs(float(g[0]) - (-6.302201890992953)) < 1e-5 * 6.3          # vs fp64 inputs: fp32 storage of P
    pn, r = ops().step_given_P(t32(k['grad_pi'][None], dev), t32(k['grad_P'][None], dev))
    assert np.allclose(pn.cpu().numpy()[0], k['grad_pi_next'], rtol=1e-6)
    assert abs(float(r[0]) - float(k['grad_reward'][0])) < 1e-5 * abs(float(k['grad_reward'][0]))


@pytest.mark.parametrize('path', sorted(glob.glob(os.path.join(G, 'functions_cfg*.npz'))))
def test_golden_functions_on_device(dev, path):
    """Reference outputs (fp64 inputs) vs the device path on fp32-rounded inputs."""
    z = np.load(path)
    d, theta, shift = int(z['d']), float(z['theta']), float(z['shift'])
    pi, P, w = z['pi'], z['P'], z['w'][:, 0]
    th = t64([theta], dev)
    a, ad = ops().alpha(t32(pi, dev), th, shift)
    pi32 = pi.astype(np.float32)
    assert rel(a.cpu().numpy(), O().calc_alpha(pi32, theta, shift), 1e-300) < 1e-9    # reference: log(1+e) (3e-12 rel. rounding at alpha~3e-5), kernel: log1p(e)
    assert np.max(np.abs(ad.cpu().numpy() - O().calc_alpha_deriv(pi32, theta, shift))) < 1e-14
    pn, r = ops().step_given_P(t32(pi, dev), t32(P, dev))
    assert np.allclose(pn.cpu().numpy(), z['pi_next'], rtol=2e-6, atol=1e-12)
    v = ops().value(t32(pi, dev), t64(w, dev)).cpu().numpy()
    assert np.allclose(v, z['value'], rtol=1e-6)
    f = ops().features(t32(pi, dev)).cpu().numpy()
    assert np.allclose(f, z['features'], rtol=1e-6, atol=1e-12)
    # reward / gradient vs the oracle on the identical (fp32-rounded) inputs: the 1e-5 bar
    P32 = P.astype(np.float32)
    assert rel(r.cpu().numpy(), O().calc_reward(P32, pi32), 1e-30) < 1e-6
    g = ops().score(t32(pi, dev), t32(P, dev), th, shift, precision='f64').cpu().numpy()
    assert rel(g, O().calc_gradient(P32, pi32, theta, shift), 1e-30) < 1e-9
    gm = ops().score(t32(pi, dev), t32(P, dev), th, shift, precision='mixed').cpu().numpy()
    assert rel(gm, O().calc_gradient(P32, pi32, theta, shift), 1e-30) < 1e-5
    # and against the reference's own fp64 outputs (input rounding included)
    assert rel(g, z['gradient'], 1e-30) < 1e-4 and rel(gm, z['gradient'], 1e-30) < 1e-4


@pytest.mark.parametrize('precision,gtol', [('f64', 1e-9), ('mixed', 1e-5)])
@pytest.mark.parametrize('d,B', [(3, 50), (4, 64), (21, 1), (21, 200), (47, 20), (64, 9), (100, 6), (128, 8), (256, 3)])
def test_td_pg_accumulate(dev, d, B, precision, gtol):
    rs = np.random.RandomState(17 + d + B)
    pi, P = rand_case(rs, B, d)
    theta, shift, gamma = 8.86349, 0.16, 0.9
    w = rs.rand(O().num_features(d))
    pn, r = ops().step_given_P(t32(pi, dev), t32(P, dev))
    delta, g, Gv = ops().td_pg_accumulate(t32(pi, dev), pn, t32(P, dev), r, t64(w, dev), t64([theta], dev), shift, gamma,
                                          precision=precision)
    pn_h = pn.cpu().numpy(); r_h = r.cpu().numpy()
    rd, rg, rGw, rGt, rsum = O().batched_td_pg(pi, pn_h, P, r_h, w, theta, shift, gamma)
    assert np.max(np.abs(delta.cpu().numpy() - rd)) < 1e-12 * max(1.0, np.max(np.abs(w)) * 4)
    assert rel(g.cpu().numpy(), rg, 1e-30) < gtol
    Gh = Gv.cpu().numpy()
    F = O().num_features(d)
    scale = np.max(np.abs(rGw)) + 1e-300
    assert np.max(np.abs(Gh[:F] - rGw)) < 1e-11 * scale
    assert abs(Gh[F] - float(np.sum(delta.cpu().numpy() * g.cpu().numpy()))) < 1e-11 * max(1.0, abs(rGt))
    assert abs(Gh[F] - rGt) < gtol * max(1.0, abs(rGt))
    assert abs(Gh[F + 1] - rsum) < 1e-9 * max(1e-12, abs(rsum)) + 1e-15
    assert Gh[F + 2] == B
    # accumulate=True adds on top
    _, _, G2 = ops().td_pg_accumulate(t32(pi, dev), pn, t32(P, dev), r, t64(w, dev), t64([theta], dev), shift, gamma,
                                      G=Gv.clone(), accumulate=True, precision=precision)
    assert np.allclose(G2.cpu().numpy(), 2 * Gh, rtol=1e-12, atol=1e-300)


def test_score_zero_probability_rule(dev):
    """P == 0 counts as 1e-100 (mfg_ac2.py:369) and P is not modified."""
    rs = np.random.RandomState(5)
    pi, P = rand_case(rs, 4, 21)
    P[:, 3, 5] = 0.0
    Pd = t32(P, dev)
    g = ops().score(t32(pi, dev), Pd, t64([8.86349], dev), 0.16, precision='f64').cpu().numpy()
    assert rel(g, O().calc_gradient(P, pi, 8.86349, 0.16), 1e-30) < 1e-9
    g = ops().score(t32(pi, dev), Pd, t64([8.86349], dev), 0.16, precision='mixed').cpu().numpy()
    assert rel(g, O().calc_gradient(P, pi, 8.86349, 0.16), 1e-30) < 1e-5
    assert np.array_equal(Pd.cpu().numpy(), P)


def test_dirichlet_from_gamma_and_gather(dev):
    rs = np.random.RandomState(9)
    y = rs.gamma(2.0, size=(5, 21, 21)).astype(np.float32)
    y[0, 0, :3] = 0.0
    P = ops().dirichlet_from_gamma(t32(y, dev)).cpu().numpy()
    assert np.allclose(P, O().dirichlet_from_gamma(y), rtol=1e-6, atol=0)
    assert P[0, 0, 0] > 0
    mat = rs.dirichlet(np.ones(21), size=64).astype(np.float32)
    idx = rs.randint(64, size=1000).astype(np.int32)
    out = ops().gather_start(t32(mat, dev), torch.as_tensor(idx, device=dev)).cpu().numpy()
    assert np.array_equal(out, mat[idx])                                  # index bookkeeping: bit exact


def test_jsd_batched(dev):
    rs = np.random.RandomState(10)
    for d in (3, 21, 128):
        p = rs.dirichlet(np.ones(d), size=50).astype(np.float32)
        q = rs.dirichlet(np.ones(d), size=50).astype(np.float32)
        p[0, 0] = 0.0
        out = ops().jsd(t32(p, dev), t32(q, dev)).cpu().numpy()
        assert np.allclose(out, O().JSD(p, q), rtol=1e-9, atol=1e-15)


# ---------------------------------------------------------------------------------------------------
# Sampler: distributional checks + invariants (RNG bit-parity with MT19937 is impossible by design)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('precision', ['mixed', 'f64'])
@pytest.mark.parametrize('d', [4, 21, 128])
def test_sampler_invariants_and_moments(dev, d, precision):
    rs = np.random.RandomState(d)
    theta, shift, scale = 8.86349, 0.16, 12000.0
    pi1 = rs.dirichlet(np.ones(d)).astype(np.float32)
    B = 4096 if d <= 21 else 1024
    pi = np.repeat(pi1[None], B, 0)
    P = ops().sample_dirichlet(t32(pi, dev), t64([theta], dev), shift, scale, seed=123, step=3,
                               precision=precision).cpu().numpy().astype(np.float64)
    assert np.all(P > 0) and np.all(np.isfinite(P))
    assert np.max(np.abs(P.sum(-1) - 1)) < 5e-7                          # rows are stochastic (test2.py:14-32)
    al = O().calc_alpha(pi1, theta, shift) * scale
    mean = al / al.sum(-1, keepdims=True)
    A = al.sum(-1, keepdims=True)
    var = mean * (1 - mean) / (A + 1)
    se = np.sqrt(var / B)
    zscore = (P.mean(0) - mean) / np.maximum(se, 1e-30)
    assert np.max(np.abs(zscore)) < 6.0
    ratio = P.var(0, ddof=1) / np.maximum(var, 1e-300)
    big = mean > 1e-3
    tol = 6.5 * np.sqrt(2.0 / (B - 1))                                     # chi-square spread of a sample variance
    assert np.all(np.abs(ratio[big] - 1) < tol)
    # determinism + counter semantics
    P2 = ops().sample_dirichlet(t32(pi, dev), t64([theta], dev), shift, scale, seed=123, step=3,
                                precision=precision).cpu().numpy()
    assert np.array_equal(P2, P.astype(np.float32))
    P3 = ops().sample_dirichlet(t32(pi, dev), t64([theta], dev), shift, scale, seed=123, step=4,
                                precision=precision).cpu().numpy()
    assert not np.array_equal(P3, P2)
    # world-size invariance: trajectories [B/2, B) drawn as a separate shard with traj_offset
    h = B // 2
    Ps = ops().sample_dirichlet(t32(pi[h:], dev), t64([theta], dev), shift, scale, seed=123, step=3,
                                traj_offset=h, precision=precision).cpu().numpy()
    assert np.array_equal(Ps, P2[h:])


def test_sampler_small_shape_regime(dev):
    """Shapes < 1 (boosted Marsaglia-Tsang) : one dominant topic makes alpha*scale ~ 0.4."""
    d = 21
    pi1 = np.zeros(d, dtype=np.float32); pi1[0] = 1.0                     # one-hot start (test2.py:63-64)
    B = 8192
    pi = np.repeat(pi1[None], B, 0)
    theta, shift, scale = 8.86349, 0.16, 12000.0
    P = ops().sample_dirichlet(t32(pi, dev), t64([theta], dev), shift, scale, seed=7).cpu().numpy().astype(np.float64)
    al = O().calc_alpha(pi1, theta, shift) * scale
    assert al.min() < 1.0
    mean = al / al.sum(-1, keepdims=True)
    var = mean * (1 - mean) / (al.sum(-1, keepdims=True) + 1)
    z = (P.mean(0) - mean) / np.sqrt(var / B)
    assert np.max(np.abs(z)) < 6.0
    assert np.all(P > 0)


# ---------------------------------------------------------------------------------------------------
# Fused rollout vs the oracle replaying the SAME sampled actions
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('d,B,T', [(3, 50, 4), (4, 37, 5), (15, 33, 6), (21, 1, 15), (21, 100, 15), (32, 10, 3), (47, 9, 4),
                                   (64, 5, 2), (65, 3, 2), (100, 5, 3), (128, 6, 3), (256, 2, 2), (320, 2, 2), (512, 1, 1),
                                   (250, 2, 2), (253, 2, 2), (450, 1, 1), (66, 3, 2), (127, 2, 2),
                                   (400, 2, 2), (448, 3, 2)])      # R = 7 (385 <= d <= 448): its register cap moved late in round 3
@pytest.mark.parametrize('discount_pow', [False, True])
@pytest.mark.parametrize('precision,gtol', [('f64', 1e-9), ('mixed', 1e-5)])
def test_rollout_fused_vs_oracle(dev, d, B, T, discount_pow, precision, gtol):
    rs = np.random.RandomState(31 + d + B)
    theta, shift, scale, gamma = 8.86349, 0.16, 12000.0, 0.9
    pi0 = rs.dirichlet(np.ones(d), size=B).astype(np.float32)
    w = rs.rand(O().num_features(d))
    out = ops().rollout(t32(pi0, dev), T, t64([theta], dev), shift, scale, w=t64(w, dev), gamma=gamma, seed=99,
                        first_step=5, traj_offset=1000, td=True, write_P=True, discount_pow=discount_pow,
                        precision=precision)
    P = out['P'].cpu().numpy()
    assert np.max(np.abs(P.astype(np.float64).sum(-1) - 1)) < 5e-7
    traj, R, D, Gs, G_w, G_theta = O().batched_rollout_given_P(
        pi0, P, w, theta, shift, gamma=gamma, variant='ac_irl' if discount_pow else 'mfg_ac2')
    assert np.allclose(out['pi_traj'].cpu().numpy(), traj, rtol=3e-7, atol=1e-12)   # <= 2 ulp fp32 per step
    # oracle replays from the kernel's own fp32 states so that per-step quantities are comparable at 1e-5
    pt = out['pi_traj'].cpu().numpy().astype(np.float64)
    r_ref = np.stack([O().calc_reward(P[:, t].astype(np.float64), pt[:, t]) for t in range(T)], 1)
    assert rel(out['reward'].cpu().numpy(), r_ref, 1e-30) < 1e-6
    g_ref = np.stack([O().calc_gradient(P[:, t], pt[:, t], theta, shift) for t in range(T)], 1)
    assert rel(out['g'].cpu().numpy(), g_ref, 1e-30) < max(gtol, 2e-7)   # ln P from ln y - ln S: fp32 P rounding
    phi = O().calc_features(pt)
    V = phi.dot(w)
    disc = gamma ** np.arange(T) if discount_pow else np.full(T, gamma)
    r_dev = out['reward'].cpu().numpy().astype(np.float64)
    d_ref = r_ref + disc[None] * V[:, 1:] - V[:, :-1]                   # the kernel bootstraps with its fp64 reward
    assert np.max(np.abs(out['delta'].cpu().numpy() - d_ref)) < 1e-11 * max(1.0, np.abs(V).max())
    # batch sums
    Gh = out['G'].cpu().numpy()
    F = O().num_features(d)
    dl = out['delta'].cpu().numpy(); gg = out['g'].cpu().numpy()
    Gw_ref = np.einsum('bt,btf->f', dl, phi[:, :T])
    assert np.max(np.abs(Gh[:F] - Gw_ref)) < 1e-11 * (np.abs(Gw_ref).max() + 1e-300)
    assert abs(Gh[F] - np.sum(dl * gg)) < 1e-10 * max(1.0, abs(np.sum(dl * gg)))
    assert abs(Gh[F + 1] - r_ref.sum()) < 2e-7 * np.abs(r_ref).sum() + 1e-18      # fp64 (in-kernel) or fp32-rounded rewards
    assert Gh[F + 2] == B * T
    # the unfused pipeline on the same actions agrees with the fused kernel
    pn, r1 = ops().step_given_P(out['pi_traj'][:, 0].contiguous(), out['P'][:, 0].contiguous())
    assert np.array_equal(pn.cpu().numpy(), out['pi_traj'][:, 1].cpu().numpy())
    assert rel(r1.cpu().numpy(), out['reward'][:, 0].cpu().numpy(), 1e-30) < 1e-6
    # sampler consistency: the standalone sampler with the same counters draws the same P
    P0 = ops().sample_dirichlet(t32(pi0, dev), t64([theta], dev), shift, scale, seed=99, step=5, traj_offset=1000,
                                precision=precision)
    assert np.array_equal(P0.cpu().numpy(), P[:, 0])


# Every instantiation R = ceil(d / 64) = 2 .. 8 of the wave-per-trajectory kernels, at a full width (d = 64 R), one column
# short of it and one column into it (d = 64 (R - 1) + 1): any edit of MFG_CORE_LARGE_WAVES* / large_row_batch (launch
# bounds, i.e. register caps and spills, and the stash size are functions of R) is covered whatever R it touches.
@pytest.mark.parametrize('R', [2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize('precision,gtol', [('f64', 1e-9), ('mixed', 1e-5)])
def test_every_wave_per_trajectory_instantiation_vs_oracle(dev, R, precision, gtol):
    theta, shift, scale, gamma = 8.86349, 0.16, 12000.0, 0.9
    for d, B, T in ((64 * R, 5, 2), (64 * R - 1, 3, 2), (64 * (R - 1) + 1, 4, 1)):
        rs = np.random.RandomState(7 * d)
        pi0 = rs.dirichlet(np.ones(d), size=B).astype(np.float32)
        w = rs.rand(O().num_features(d))
        out = ops().rollout(t32(pi0, dev), T, t64([theta], dev), shift, scale, w=t64(w, dev), gamma=gamma, seed=123,
                            first_step=2, traj_offset=77, td=True, write_P=True, precision=precision)
        P = out['P'].cpu().numpy()
        assert np.isfinite(P).all() and np.max(np.abs(P.astype(np.float64).sum(-1) - 1)) < 5e-7
        pt = out['pi_traj'].cpu().numpy().astype(np.float64)
        traj = O().batched_rollout_given_P(pi0, P, w, theta, shift, gamma=gamma)[0]
        assert np.allclose(pt, traj, rtol=3e-7, atol=1e-12)
        r_ref = np.stack([O().calc_reward(P[:, t].astype(np.float64), pt[:, t]) for t in range(T)], 1)
        assert rel(out['reward'].cpu().numpy(), r_ref, 1e-30) < 1e-6
        g_ref = np.stack([O().calc_gradient(P[:, t], pt[:, t], theta, shift) for t in range(T)], 1)
        assert rel(out['g'].cpu().numpy(), g_ref, 1e-30) < max(gtol, 2e-7)
        V = O().calc_features(pt).dot(w)
        d_ref = r_ref + gamma * V[:, 1:] - V[:, :-1]
        assert np.max(np.abs(out['delta'].cpu().numpy() - d_ref)) < 1e-11 * max(1.0, np.abs(V).max())
        # the sampling-only instantiation (no TD: another register budget) draws the same actions
        env = ops().rollout(t32(pi0, dev), T, t64([theta], dev), shift, scale, seed=123, first_step=2, traj_offset=77, td=False,
                            write_P=True, precision=precision)
        assert np.array_equal(env['P'].cpu().numpy(), P)
        P0 = ops().sample_dirichlet(t32(pi0, dev), t64([theta], dev), shift, scale, seed=123, step=2, traj_offset=77,
                                    precision=precision)
        assert np.array_equal(P0.cpu().numpy(), P[:, 0])


def test_rollout_env_only_matches_td_rollout(dev):
    rs = np.random.RandomState(77)
    d, B, T = 21, 64, 15
    pi0 = rs.dirichlet(np.ones(d), size=B).astype(np.float32)
    th = t64([8.86349], dev)
    w = t64(rs.rand(O().num_features(d)), dev)
    a = ops().rollout(t32(pi0, dev), T, th, 0.16, 12000.0, w=w, seed=1, td=True)
    b = ops().rollout(t32(pi0, dev), T, th, 0.16, 12000.0, seed=1, td=False)
    assert np.array_equal(a['pi_traj'].cpu().numpy(), b['pi_traj'].cpu().numpy())
    assert np.array_equal(a['reward'].cpu().numpy(), b['reward'].cpu().numpy())


def test_apply_update_batch1_equals_reference_increment(dev):
    """B = 1: G / count is exactly (delta*phi, delta*g), so the update is the reference's (mfg_ac2.py:511-522)."""
    z = np.load(os.path.join(G, 'train_mfg_ac2_c1_g1.npz'))
    pi, P = z['pi'][0][None], z['P'][0][None]
    w0 = z['w0'][:, 0].copy(); theta0 = float(z['theta0'])
    th = t64([theta0], dev); w = t64(w0, dev)
    pn, r = ops().step_given_P(t32(pi, dev), t32(P, dev))
    delta, g, Gv = ops().td_pg_accumulate(t32(pi, dev), pn, t32(P, dev), r, w, th, float(z['shift']), 1.0, precision='f64')
    ops().apply_update(Gv, 21, 0.1, 0.001, w, th)
    # reference: theta after the first step of the constant-lr trace
    assert abs(float(th[0]) - float(z['theta_before'][1])) < 2e-6 * abs(float(z['theta_before'][1]) - theta0) + 1e-12
    # oracle on the identical fp32 inputs (pi, P, the kernel's fp32 pi' and reward)
    pi32, P32 = pi.astype(np.float32), P.astype(np.float32)
    rd, rg, rGw, rGt, _ = O().batched_td_pg(pi32, pn.cpu().numpy(), P32, r.cpu().numpy(), w0, theta0,
                                            float(z['shift']), 1.0)
    assert abs(float(th[0]) - (theta0 + 0.001 * rGt)) < 1e-12
    assert np.max(np.abs(w.cpu().numpy() - (w0 + 0.1 * rGw))) < 1e-13


def test_error_paths(dev):
    from discrete_mean_field_game_amd import _lib as L
    pi = torch.zeros(2, 600, device=dev)
    P = torch.zeros(2, 600, 600, device=dev)
    with pytest.raises(L.MfgError):
        ops().step_given_P(pi, P)                                         # d > MFG_MAX_D
    # B = 0 is a no-op
    pn, r = ops().step_given_P(torch.zeros(0, 21, device=dev), torch.zeros(0, 21, 21, device=dev))
    assert pn.shape == (0, 21) and r.shape == (0,)


@pytest.mark.parametrize('theta', [0.5, 27.0, 40.0, 60.0])
def test_score_beyond_the_h_table_and_extreme_theta(dev, theta):
    """Mixed precision evaluates -psi(alpha) alpha' from the h(z) table for |z| < 24 and falls back to the direct
    form above it (theta (1 - shift) > 24); both must stay within the 1e-5 bar.  At large theta the reference's
    own softplus, log(1 + exp(z)) (mfg_ac2.py:228), loses all digits of alpha for z < -36 (measured 3e-4 on g at
    theta = 40), so the yardstick here is the same formula with log1p."""
    rs = np.random.RandomState(int(theta))
    pi, P = rand_case(rs, 64, 21, conc=0.3)
    shift = 0.16
    z = theta * (pi.astype(np.float64)[:, None, :] - pi.astype(np.float64)[:, :, None] - shift)
    ref = O().calc_gradient(P, pi, theta, shift, mat_alpha=np.log1p(np.exp(z)))
    gm = ops().score(t32(pi, dev), t32(P, dev), t64([theta], dev), shift, precision='mixed').cpu().numpy()
    gf = ops().score(t32(pi, dev), t32(P, dev), t64([theta], dev), shift, precision='f64').cpu().numpy()
    assert np.all(np.isfinite(gm)) and np.all(np.isfinite(gf))
    assert rel(gf, ref, 1e-30) < 1e-9
    assert rel(gm, ref, 1e-30) < 1e-5
    if theta < 20:                                            # where the reference formula is still accurate
        assert rel(gf, O().calc_gradient(P, pi, theta, shift), 1e-30) < 1e-9


@pytest.mark.parametrize('d,N', [(4, 9), (15, 30), (21, 64), (100, 3)])
def test_policy_logpdf_vs_oracle(dev, d, N):
    """f1: log-density of the product-Dirichlet policy under K policies (ac_irl.py:270-289, :324-379)."""
    rs = np.random.RandomState(d)
    pi, P = rand_case(rs, N, d, conc=1.0, pconc=2.0)
    thetas = np.array([2.0, 6.5, 8.64])
    for scale, floor, pfloor in [(1.0, 0.0, 0.0), (1.0, 1.0 + 1e-6, 0.0), (50.0, 0.0, 1e-6)]:
        got = ops().policy_logpdf(t32(pi, dev), t32(P, dev), t64(thetas, dev), 0.05, scale, floor, pfloor).cpu().numpy()
        want = O().policy_logpdf(pi, P, thetas, 0.05, scale, floor, pfloor)
        assert got.shape == (N, 3)
        assert rel(got, want, floor=1.0) < 1e-10
    # an exact zero in an action: density 0 (log = -inf) when alpha > 1, like tf.distributions.Dirichlet.prob
    P0 = P.copy()
    P0[0, 1, 2] = 0.0
    got = ops().policy_logpdf(t32(pi, dev), t32(P0, dev), t64(thetas, dev), 0.05, 1.0, 1.0 + 1e-6).cpu().numpy()
    assert np.all(np.isneginf(got[0])) and np.all(np.isfinite(got[1:]))


@pytest.mark.parametrize('d,B', [(21, 50), (15, 37), (64, 5), (100, 4), (128, 6), (256, 3), (130, 3)])
@pytest.mark.parametrize('off', [1, 2])
def test_step_given_P_misaligned_pointers(dev, d, B, off):
    """Buffers that are only 4- or 8-byte aligned (views into a larger allocation) take the scalar / narrow-vector
    fallbacks and must give exactly what the 16-byte-aligned fast paths give."""
    rs = np.random.RandomState(100 + d)
    pi, P = rand_case(rs, B, d)
    pa, Pa = t32(pi, dev), t32(P, dev)
    want_pi, want_r = ops().step_given_P(pa, Pa)
    bufP = torch.zeros(B * d * d + 8, device=dev)
    bufpi = torch.zeros(B * d + 8, device=dev)
    Pv = bufP[off:off + B * d * d].view(B, d, d)
    pv = bufpi[off:off + B * d].view(B, d)
    Pv.copy_(Pa)
    pv.copy_(pa)
    assert Pv.data_ptr() % 16 != 0
    got_pi, got_r = ops().step_given_P(pv, Pv)
    assert torch.equal(got_pi, want_pi)
    assert rel(got_r.cpu().numpy(), want_r.cpu().numpy(), floor=1e-3) < 1e-6
    ref_pi = O().transition(P.astype(np.float64), pi.astype(np.float64))
    np.testing.assert_allclose(got_pi.cpu().numpy(), ref_pi, rtol=0, atol=1e-7)


@pytest.mark.parametrize('precision', ['mixed', 'f64'])
@pytest.mark.parametrize('first_step', [0, 1, 6, 7])
@pytest.mark.parametrize('d,B', [(21, 300), (5, 257), (15, 260), (9, 64)])
def test_rollout_does_not_depend_on_how_it_is_cut_into_launches(dev, d, B, first_step, precision):
    """A T-step sampled rollout equals the chain of T one-step launches fed forward (first_step + s), bit for bit, from even
    AND odd first steps.  Rows of length 1 mod 4 (d = 21, 5, 9) draw their trailing element from a Box-Muller pair keyed by
    the EVEN step and carry its partner to the odd step behind it (sample_tail1): a launch that starts on an odd step has
    nothing carried and must recompute the pair."""
    o_ = ops()
    rs = np.random.RandomState(d + B + first_step)
    pi = t32(rs.dirichlet(np.ones(d), size=B), dev)
    th = t64([8.64], dev)
    T = 5
    whole = o_.rollout(pi, T, th, 0.16, 12000.0, seed=11, first_step=first_step, traj_offset=5, td=False, write_P=True,
                       precision=precision)
    Pw = whole['P'].view(B, T, d, d)
    cur = pi
    for s in range(T):
        one = o_.rollout(cur, 1, th, 0.16, 12000.0, seed=11, first_step=first_step + s, traj_offset=5, td=False, write_P=True,
                         precision=precision)
        assert torch.equal(one['P'].view(B, d, d), Pw[:, s]), (s, first_step)
        cur = one['pi_last'].contiguous()
    assert torch.equal(cur, whole['pi_last'])
    # ... and a cut in the middle, at an odd and at an even step
    for cut in (2, 3):
        a1 = o_.rollout(pi, cut, th, 0.16, 12000.0, seed=11, first_step=first_step, traj_offset=5, td=False, write_P=True,
                        precision=precision)
        a2 = o_.rollout(a1['pi_last'].contiguous(), T - cut, th, 0.16, 12000.0, seed=11, first_step=first_step + cut,
                        traj_offset=5, td=False, write_P=True, precision=precision)
        assert torch.equal(a1['P'].view(B, cut, d, d), Pw[:, :cut]) and torch.equal(a2['P'].view(B, T - cut, d, d), Pw[:, cut:])


@pytest.mark.parametrize('d,B', [(21, 100), (15, 64), (47, 9), (100, 5), (4, 33)])
@pytest.mark.parametrize('precision,gtol', [('f64', 1e-9), ('mixed', 1e-5)])
def test_external_reward_step_equals_two_pass_step(dev, d, B, precision, gtol):
    """The IRL step (ac_irl.py:679-708): rollout(T=1, reward_kind=EXTERNAL) leaves delta = disc*V(pi') - V(pi) and g,
    mfg_grad_accumulate(add_reward) folds the network's reward in -- same actions, same sums as sampling P first and
    running mfg_td_pg_accumulate on it afterwards, and both agree with the oracle on the sampled actions."""
    from discrete_mean_field_game_amd import _lib as L
    o_ = ops()
    rs = np.random.RandomState(d + B)
    pi = t32(rs.dirichlet(np.ones(d), size=B), dev)
    w = t64(rs.rand(o_.num_features(d)), dev)
    th = t64([8.64], dev)
    disc = 0.81
    a = o_.rollout(pi, 1, th, 0.0, 1e4, seed=9, first_step=4, td=False, write_P=True, precision=precision)
    P = a['P'].view(B, d, d)
    r = torch.tanh(5.0 * torch.einsum('bi,bii->b', pi, P) - 0.3).contiguous()           # any reward of (pi, P)
    ws = o_.workspace(B, d, dev)
    Ga = torch.zeros(o_.num_features(d) + 3, dtype=torch.float64, device=dev)
    da, ga, _ = o_.td_pg_accumulate(pi, a['pi_last'], P, r, w, th, 0.0, disc, G=Ga, ws=ws, precision=precision)
    b = o_.rollout(pi, 1, th, 0.0, 1e4, w=w, gamma=disc, reward_kind=L.REWARD_EXTERNAL, seed=9, first_step=4, td=True,
                   write_P=True, precision=precision)
    assert b['reward'] is None and torch.equal(b['P'], a['P']) and torch.equal(b['pi_last'], a['pi_last'])
    Gb = torch.zeros_like(Ga)
    delta = b['delta'].view(B).clone()
    o_.grad_accumulate(pi, delta, b['g'].view(B), r, Gb, ws, add_reward=True)
    assert rel(delta.cpu().numpy(), da.cpu().numpy(), floor=1e-3) < 1e-12
    gt = max(gtol, 2e-7)      # fused: ln P = ln y - ln S; two-pass: ln of the fp32-rounded P
    assert rel(b['g'].view(B).cpu().numpy(), ga.cpu().numpy(), floor=1.0) < gt
    Q = o_.num_features(d)
    assert rel(Gb[:Q].cpu().numpy(), Ga[:Q].cpu().numpy(), floor=float(Ga[:Q].abs().max())) < 1e-12
    assert abs(float(Gb[Q]) - float(Ga[Q])) <= gt * max(1.0, float((da * ga).abs().sum()))
    assert abs(float(Gb[Q + 1]) - float(Ga[Q + 1])) <= 1e-12 * max(1.0, abs(float(Ga[Q + 1])))
    assert float(Gb[Q + 2]) == B
    # oracle on the sampled actions
    dl, gg, G_w, G_t, _ = O().batched_td_pg(pi.cpu().numpy(), a['pi_last'].cpu().numpy(), P.cpu().numpy().astype(np.float64),
                                            r.cpu().numpy().astype(np.float64), w.cpu().numpy(), 8.64, 0.0, disc)
    assert rel(delta.cpu().numpy(), dl, floor=1e-2) < 1e-5
    assert rel(Gb[:Q].cpu().numpy(), G_w, floor=float(np.abs(G_w).max())) < 1e-5
    # accumulate=True adds a second identical batch
    o_.grad_accumulate(pi, b['delta'].view(B).clone(), b['g'].view(B), r, Gb, ws, add_reward=True, accumulate=True)
    assert rel(Gb[:Q].cpu().numpy(), 2 * Ga[:Q].cpu().numpy(), floor=float(Ga[:Q].abs().max())) < 1e-12
    # misuse is refused
    with pytest.raises(Exception):
        o_.rollout(pi, 1, th, 0.0, 1e4, w=w, reward_kind=L.REWARD_EXTERNAL, td=True, G=Ga, ws=ws)


def test_one_workspace_serves_calls_of_different_sizes(dev):
    """A workspace sized for the largest batch is reused by smaller calls (a 15-step rollout, then single steps):
    the completion counter of the in-kernel finalisation lives at a fixed place, not behind the rows of one N."""
    o_ = ops()
    d, B, T = 21, 3000, 15
    rs = np.random.RandomState(1)
    pi = t32(rs.dirichlet(np.ones(d), size=B), dev)
    w = t64(rs.rand(o_.num_features(d)), dev)
    th = t64([8.86349], dev)
    ws = o_.workspace(B * T, d, dev)
    big = o_.rollout(pi, T, th, 0.16, 12000.0, w=w, seed=1, td=True, ws=ws)
    for n in (B, 7, 500, 64):
        sub = pi[:n].contiguous()
        a = o_.rollout(sub, 1, th, 0.16, 12000.0, w=w, seed=2, td=True, ws=ws)                 # shared, oversized
        b = o_.rollout(sub, 1, th, 0.16, 12000.0, w=w, seed=2, td=True)                        # its own workspace
        assert torch.equal(a['G'], b['G']) and float(a['G'][-1]) == n
    again = o_.rollout(pi, T, th, 0.16, 12000.0, w=w, seed=1, td=True, ws=ws)
    assert torch.equal(again['G'], big['G'])


@pytest.mark.parametrize('d', [1, 2, 63, 511, 512])
def test_extreme_dimensions_against_oracle(dev, d):
    """Smallest and largest supported d (MFG_MAX_D = 512) through every stage: given-P step, TD/score, fused rollout."""
    o_ = ops()
    B, T = (3, 2) if d >= 511 else (7, 3)
    rs = np.random.RandomState(d)
    pi, P = rand_case(rs, B, d)
    w = rs.rand(O().num_features(d))
    theta, shift, scale = 4.0, 0.05, 500.0
    pn, r = o_.step_given_P(t32(pi, dev), t32(P, dev))
    np.testing.assert_allclose(pn.cpu().numpy(), O().transition(P.astype(np.float64), pi.astype(np.float64)), rtol=0, atol=2e-7)
    r_ref = O().calc_reward(P.astype(np.float64), pi.astype(np.float64))
    assert rel(r.cpu().numpy(), r_ref, floor=1e-6) < 1e-5
    for precision, gtol in (('f64', 1e-9), ('mixed', 1e-5)):
        dl, gg, G = o_.td_pg_accumulate(t32(pi, dev), pn, t32(P, dev), r, t64(w, dev), t64([theta], dev), shift, 0.9,
                                        precision=precision)
        d_ref, g_ref, G_w, G_t, _ = O().batched_td_pg(pi, pn.cpu().numpy(), P, r.cpu().numpy().astype(np.float64), w, theta,
                                                      shift, 0.9)
        assert rel(dl.cpu().numpy(), d_ref, floor=1e-2) < 1e-9
        assert rel(gg.cpu().numpy(), g_ref, floor=1.0) < gtol
        F = O().num_features(d)
        assert rel(G[:F].cpu().numpy(), G_w, floor=float(np.abs(G_w).max()) + 1e-300) < 1e-9
    out = o_.rollout(t32(pi, dev), T, t64([theta], dev), shift, scale, w=t64(w, dev), gamma=0.9, seed=5, td=True, write_P=True)
    Pk = out['P'].cpu().numpy()
    assert np.max(np.abs(Pk.astype(np.float64).sum(-1) - 1)) < 2e-6 and Pk.min() >= 0
    traj, R, D, Gs, _, _ = O().batched_rollout_given_P(pi, Pk, w, theta, shift, gamma=0.9)
    np.testing.assert_allclose(out['pi_traj'].cpu().numpy(), traj, rtol=0, atol=3e-7)
    assert rel(out['reward'].cpu().numpy(), R, floor=1e-6) < 1e-4
    assert rel(out['delta'].cpu().numpy(), D, floor=1e-2) < 1e-5
    assert rel(out['g'].cpu().numpy(), Gs, floor=1.0) < 1e-5


def test_empty_batches_and_bad_arguments(dev):
    """B = 0 is a no-op everywhere; undersized workspaces, null pointers and d > 512 are refused with an error code
    and a message instead of a launch."""
    from discrete_mean_field_game_amd import _lib as L
    o_ = ops()
    d = 21
    th = t64([8.0], dev)
    w = t64(np.zeros(o_.num_features(d)), dev)
    e2 = torch.zeros(0, d, device=dev)
    e3 = torch.zeros(0, d, d, device=dev)
    assert o_.sample_dirichlet(e2, th, 0.1, 100.0, seed=1).shape == (0, d, d)
    out = o_.rollout(e2, 3, th, 0.1, 100.0, w=w, seed=1, td=True)
    assert out['pi_traj'].shape == (0, 4, d) and out['delta'].shape == (0, 3)
    assert o_.jsd(e2, e2).shape == (0,)
    assert o_.value(e2, w).shape == (0,)
    lib = L.lib()
    pi = torch.rand(8, d, device=dev)
    P = torch.rand(8, d, d, device=dev)
    buf = torch.zeros(8, d, device=dev)
    r = torch.zeros(8, device=dev)
    dl = torch.zeros(8, dtype=torch.float64, device=dev)
    G = torch.zeros(o_.num_features(d) + 3, dtype=torch.float64, device=dev)
    small = torch.zeros(4, dtype=torch.float64, device=dev)
    rc = lib.mfg_td_pg_accumulate(pi.data_ptr(), buf.data_ptr(), P.data_ptr(), r.data_ptr(), w.data_ptr(), th.data_ptr(), 0.1,
                                  1.0, 8, d, L.PRECISION_MIXED, dl.data_ptr(), dl.data_ptr(), G.data_ptr(), 0,
                                  small.data_ptr(), small.numel() * 8, None)
    assert rc == -4 and b'workspace' in lib.mfg_last_error()                     # MFG_EWORKSPACE
    assert lib.mfg_step_given_P(None, P.data_ptr(), 8, d, 0, buf.data_ptr(), r.data_ptr(), None) == -1   # MFG_EINVAL
    assert lib.mfg_step_given_P(pi.data_ptr(), P.data_ptr(), 8, 513, 0, buf.data_ptr(), r.data_ptr(), None) == -3
    assert lib.mfg_step_given_P(pi.data_ptr(), P.data_ptr(), 8, d, 7, buf.data_ptr(), r.data_ptr(), None) == -1
    assert lib.mfg_rollout(pi.data_ptr(), 8, d, 0, th.data_ptr(), 0.1, 100.0, None, 1.0, 0, 1, 0, 0, 0, None, None, None, None,
                           None, None, None, 0, None, 0, None) == -1             # T < 1
    torch.cuda.synchronize()                                                     # nothing was launched, nothing is broken
    pn, _ = o_.step_given_P(pi, P)
    assert torch.isfinite(pn).all()


def _sweep_cases():
    rs = np.random.RandomState(2024)
    cases = []
    for _ in range(36):
        cases.append((int(rs.randint(1, 65)), int(rs.randint(1, 70)), int(rs.randint(1, 5))))
    for _ in range(8):
        cases.append((int(rs.randint(65, 400)), int(rs.randint(1, 9)), int(rs.randint(1, 3))))
    return cases


@pytest.mark.parametrize('d,B,T', _sweep_cases())
def test_random_shape_sweep(dev, d, B, T):
    """Seeded sweep over (d, B, T): every packing (G = 64/d trajectories per wave, odd/even d, ragged last tile,
    1..7 columns per lane above d = 64) through the fused rollout, the given-P kernel and the gradient sums."""
    o_ = ops()
    rs = np.random.RandomState(1000 * d + B)
    theta, shift, scale, gamma = float(rs.uniform(1, 12)), float(rs.uniform(0, 0.3)), float(10 ** rs.uniform(1, 4.5)), 0.93
    pi0 = rs.dirichlet(np.ones(d) * rs.choice([0.3, 1.0, 5.0]), size=B).astype(np.float32)
    w = rs.rand(O().num_features(d))
    precision = 'mixed' if (d + B) % 2 else 'f64'
    out = o_.rollout(t32(pi0, dev), T, t64([theta], dev), shift, scale, w=t64(w, dev), gamma=gamma, seed=d * 7 + B,
                     first_step=3, traj_offset=11, td=True, write_P=True, precision=precision)
    P = out['P'].cpu().numpy()
    assert np.isfinite(P).all() and P.min() >= 0 and np.max(np.abs(P.astype(np.float64).sum(-1) - 1)) < 2e-6
    traj, R, D, Gs, G_w, G_t = O().batched_rollout_given_P(pi0, P, w, theta, shift, gamma=gamma)
    np.testing.assert_allclose(out['pi_traj'].cpu().numpy(), traj, rtol=0, atol=3e-7)
    pt = out['pi_traj'].cpu().numpy().astype(np.float64)
    r_ref = np.stack([O().calc_reward(P[:, t].astype(np.float64), pt[:, t]) for t in range(T)], 1)
    assert rel(out['reward'].cpu().numpy(), r_ref, floor=1e-6) < 1e-5
    g_ref = np.stack([O().calc_gradient(P[:, t], pt[:, t], theta, shift) for t in range(T)], 1)
    assert rel(out['g'].cpu().numpy(), g_ref, floor=1.0) < 1e-5
    V = O().calc_features(pt).dot(w)
    d_ref = out['reward'].cpu().numpy().astype(np.float64) + gamma * V[:, 1:] - V[:, :-1]
    assert np.max(np.abs(out['delta'].cpu().numpy() - d_ref)) < 1e-6 * max(1.0, np.abs(V).max())
    dl, gg = out['delta'].cpu().numpy(), out['g'].cpu().numpy()
    F = O().num_features(d)
    Gw_ref = np.einsum('bt,btf->f', dl, O().calc_features(pt)[:, :T])
    Gh = out['G'].cpu().numpy()
    assert np.max(np.abs(Gh[:F] - Gw_ref)) <= 1e-10 * (np.abs(Gw_ref).max() + 1e-300)
    assert abs(Gh[F] - np.sum(dl * gg)) <= 1e-9 * max(1.0, np.abs(dl * gg).sum())
    assert Gh[F + 2] == B * T
    pn, r1 = o_.step_given_P(out['pi_traj'][:, 0].contiguous(), out['P'][:, 0].contiguous())
    assert torch.equal(pn, out['pi_traj'][:, 1])
    assert rel(r1.cpu().numpy(), out['reward'][:, 0].cpu().numpy(), floor=1e-6) < 2e-6


@pytest.mark.parametrize('d,scale,theta', [(4, 3.0, 2.0), (4, 0.7, 2.0), (21, 12000.0, 8.86349), (21, 30.0, 6.0), (130, 200.0, 5.0)])
def test_sampler_marginals_are_beta_distributed(dev, d, scale, theta):
    """Distributional parity of the in-kernel Dirichlet sampler (mfg_ac2.py:236-254): the marginal of P_ij is
    Beta(a_ij, A_i - a_ij).  Kolmogorov-Smirnov on 20 000 draws per checked entry, shapes from << 1 (boosted
    Marsaglia-Tsang) to ~1e4; seeded, so the p-value bound is not a flake source."""
    from scipy import stats
    rs = np.random.RandomState(7 * d)
    pi1 = rs.dirichlet(np.ones(d)).astype(np.float32)
    B = 20000
    pi = np.repeat(pi1[None], B, 0)
    P = ops().sample_dirichlet(t32(pi, dev), t64([theta], dev), 0.1, scale, seed=2025, step=1).cpu().numpy().astype(np.float64)
    al = O().calc_alpha(pi1, theta, 0.1) * scale
    A = al.sum(-1)
    pmin = 1.0
    for (i, j) in [(0, 0), (0, d - 1), (d // 2, 1), (d - 1, d // 2), (1, d - 2)]:
        a, b = al[i, j], A[i] - al[i, j]
        ks = stats.kstest(P[:, i, j], stats.beta(a, b).cdf)
        pmin = min(pmin, ks.pvalue)
        assert ks.pvalue > 1e-4, (i, j, a, b, ks)
    # independence across rows: correlation of entries of different rows is sampling noise
    c = np.corrcoef(P[:, 0, 0], P[:, 1, 0])[0, 1]
    assert abs(c) < 5.0 / np.sqrt(B)


@pytest.mark.parametrize('d,B,T', [(64, 37, 3), (80, 21, 2), (96, 50, 1), (128, 333, 2), (144, 9, 1), (256, 70, 2), (512, 5, 1),
                                   (128, 9001, 3), (256, 4403, 2), (128, 1, 1), (256, 1, 1),
                                   (21, 1000, 4), (15, 77, 3), (47, 40, 2), (3, 50, 4), (4, 1, 1), (21, 1, 1), (21, 3, 1),
                                   (28, 33, 2), (29, 10, 2), (17, 129, 5), (21, 20000, 15), (21, 4096, 1), (15, 8192, 15)])
@pytest.mark.parametrize('add_reward', [False, True])
def test_grad_accumulate_all_kernels(dev, d, B, T, add_reward):
    """The batch sums on their own (a6/a8) for every gradient kernel: fp64-MFMA tiles (d multiple of 16, >= 64; d = 80
    and 144 take its scalar staging path; d = 128 / 256 the register-blocked k_grad_mfma2 -- the 27 003- and 8 806-sample
    cases give its blocks several chunks each, the one-sample cases a single ragged one), the augmented-vector fp64-MFMA kernel of d <= 28 (round 3; compile-time
    d = 21 / 15 and run-time d; one to 1 024 partial rows, in-kernel and separate finalisation), the generic one (d = 29,
    47); trajectory-major layout with stride (T+1) d, ragged sample counts, optional delta += reward."""
    o_ = ops()
    rs = np.random.RandomState(d * 3 + B)
    traj = t32(rs.dirichlet(np.ones(d), size=(B, T + 1)), dev)                 # [B, T+1, d]
    N = B * T
    delta = t64(rs.randn(B, T), dev)
    g = t64(rs.randn(B, T), dev)
    r = t32(rs.rand(B, T), dev)
    F = o_.num_features(d)
    G = torch.full((F + 3,), 7.0, dtype=torch.float64, device=dev)
    ws = o_.workspace(N, d, dev)
    dl = delta.clone()
    o_.grad_accumulate(traj, dl, g, r, G, ws, T=T, add_reward=add_reward)
    de = (delta + r.double()) if add_reward else delta
    assert torch.equal(dl, de)
    x = traj[:, :T].double().reshape(N, d)
    dv = de.reshape(N)
    M = (x * dv[:, None]).T @ x
    iu = torch.triu_indices(d, d, device=dev)
    Q = d * (d + 1) // 2
    ref = torch.cat([M[iu[0], iu[1]], (x * dv[:, None]).sum(0), dv.sum()[None], (dv * g.reshape(N)).sum()[None],
                     r.double().sum()[None], torch.tensor([float(N)], dtype=torch.float64, device=dev)])
    scale = float(ref[:Q].abs().max())
    assert float((G[:Q] - ref[:Q]).abs().max()) <= 1e-12 * max(scale, 1e-300)
    assert float((G[Q:] - ref[Q:]).abs().max()) <= 1e-11 * max(1.0, float(ref[Q:].abs().max()))
    # accumulate = True adds onto G
    o_.grad_accumulate(traj, de.clone(), g, r, G, ws, T=T, accumulate=True)
    assert float((G[:Q] - 2 * ref[:Q]).abs().max()) <= 1e-12 * max(scale, 1e-300)


@pytest.mark.parametrize('d,B', [(21, 1), (21, 12), (21, 13), (21, 4096), (21, 6144), (21, 6145), (15, 16), (15, 5000)])
@pytest.mark.parametrize('precision', ['mixed', 'f64'])
def test_step_kernel_fused_batch_sums(dev, d, B, precision):
    """Per-step updates (T = 1) at the packed sizes: the step kernel itself leaves the batch sums of its tile
    (k_core_small<..., SUMS>, fp64 MFMA on the augmented vectors) and a row reduction finishes them; above 512 tiles the
    two-kernel path takes over.  G must equal the sums formed by the stand-alone gradient kernel over the SAME outputs
    (re-associated fp64 sums: 1e-12), both paths must produce identical per-trajectory outputs, and `accumulate` must
    add onto G."""
    o_ = ops()
    rs = np.random.RandomState(d + B)
    pi0 = t32(rs.dirichlet(np.ones(d), size=B), dev)
    F = o_.num_features(d)
    w = t64(rs.rand(F), dev)
    th = t64([8.86349], dev)
    ws = o_.workspace(B, d, dev)
    out = o_.rollout(pi0, 1, th, 0.16, 12000.0, w=w, gamma=0.9, seed=3, first_step=7, td=True, ws=ws, precision=precision)
    G = out['G'].clone()
    ref = torch.zeros_like(G)
    o_.grad_accumulate(out['pi_traj'], out['delta'].view(-1).clone(), out['g'].view(-1), out['reward'].view(-1), ref,
                       o_.workspace(B, d, dev), T=1)
    scale = float(ref[:F].abs().max())
    assert float((G[:F] - ref[:F]).abs().max()) <= 1e-12 * max(scale, 1e-300)
    assert float((G[F:] - ref[F:]).abs().max()) <= 1e-11 * max(1.0, float(ref[F:].abs().max()))
    assert float(G[F + 2]) == B
    # the same launch without batch sums (no G): identical per-trajectory outputs
    out2 = o_.rollout(pi0, 1, th, 0.16, 12000.0, w=w, gamma=0.9, seed=3, first_step=7, td=True, reward_kind=2, precision=precision)
    assert torch.equal(out2['pi_traj'], out['pi_traj']) and torch.equal(out2['g'], out['g'])
    o_.rollout(pi0, 1, th, 0.16, 12000.0, w=w, gamma=0.9, seed=3, first_step=7, td=True, ws=ws, G=G, accumulate=True,
               precision=precision)
    assert float((G[:F] - 2 * ref[:F]).abs().max()) <= 1e-12 * max(scale, 1e-300) and float(G[F + 2]) == 2 * B


def test_mixed_sampler_range_of_the_separable_exponential(dev):
    """Mixed precision forms e^{theta (pi_j - pi_i - shift)} as E_j F_i, fp32 factors centred on pi = 1/2.  Inside the
    documented range, |theta| (1/2 + |shift|) <= 86, the sampler is still exact (KS on Beta marginals at theta = 40 and,
    on a peaked state, at theta = 100, where the uncentred factors of round 2 overflowed).  Beyond it the launch must not
    fail silently: the outputs are NaN, the device status word reports MFG_STATUS_MIXED_RANGE, and every later
    mixed-precision sampling launch is refused with MFG_ERANGE until mfg_clear_status(); precision 'f64' has no such limit."""
    from scipy import stats
    from discrete_mean_field_game_amd import _lib as L
    o_ = ops()
    o_.clear_status()
    d, B = 6, 20000
    pi1 = np.array([0.05, 0.3, 0.1, 0.35, 0.15, 0.05], dtype=np.float32)
    pi = np.repeat(pi1[None], B, 0)
    shift, scale = 0.1, 50.0
    for theta, pi1 in ((40.0, pi1), (100.0, np.array([0.97, 0.006, 0.006, 0.006, 0.006, 0.006], dtype=np.float32))):
        pi = np.repeat(pi1[None], B, 0)
        P = o_.sample_dirichlet(t32(pi, dev), t64([theta], dev), shift, scale, seed=11, precision='mixed').cpu().numpy().astype(np.float64)
        assert np.all(np.isfinite(P)) and np.allclose(P.sum(-1), 1.0, atol=1e-5)
        al = O().calc_alpha(pi1.astype(np.float64), theta, shift) * scale
        for (i, j) in [(0, 3), (3, 0), (1, 3), (2, 2), (1, 0)]:
            a, b = al[i, j], al[i].sum() - al[i, j]
            if a < 0.5 or b < 0.5 or not np.isfinite(a + b):
                continue                                          # numerically a point mass at 0 / 1 (fp32 storage saturates)
            ks = stats.kstest(P[:, i, j], stats.beta(a, b).cdf)
            assert ks.pvalue > 1e-4, (theta, i, j, a, b, ks)
    assert o_.status() == 0
    # |theta| (1/2 + |shift|) = 120 > 86: out of range
    big = t64([200.0], dev)
    peaked = np.repeat(np.array([[0.97, 0.006, 0.006, 0.006, 0.006, 0.006]], dtype=np.float32), 64, 0)
    Pm = o_.sample_dirichlet(t32(peaked, dev), big, shift, scale, seed=11, precision='mixed')
    assert not bool(torch.isfinite(Pm).all())                     # NaN, not a silently wrong sample
    assert o_.status() == L.STATUS_MIXED_RANGE
    with pytest.raises(L.MfgError, match='86'):                   # sticky: the next policy launch is refused
        o_.sample_dirichlet(t32(peaked, dev), t64([8.0], dev), shift, scale, seed=11, precision='mixed')
    # ... and only the launches the condition concerns: strict-precision sampling and launches on given actions (another
    # model instance or thread on this device) are not held up by it
    Pf = o_.sample_dirichlet(t32(peaked, dev), big, shift, scale, seed=11, precision='f64')
    assert bool(torch.isfinite(Pf).all()) and bool((Pf.sum(-1) - 1).abs().max() < 1e-5)
    assert bool(torch.isfinite(o_.score(t32(peaked, dev), Pf, t64([8.0], dev), shift)).all())
    assert o_.status() == L.STATUS_MIXED_RANGE                   # still set: f64 launches neither report nor clear it
    o_.clear_status()
    assert o_.status() == 0
    # a NaN theta is reported as well (the test is !(x <= limit))
    o_.sample_dirichlet(t32(peaked, dev), t64([float('nan')], dev), shift, scale, seed=11, precision='mixed')
    assert o_.status() == L.STATUS_MIXED_RANGE
    o_.clear_status()


def test_diverging_training_run_stops_with_an_error(dev):
    """theta lives on the device: a mixed-precision run whose theta is driven out of the separable exponential's range
    must end in MfgError (MFG_ERANGE) on a later call instead of training on NaNs."""
    from discrete_mean_field_game_amd import _lib as L
    from discrete_mean_field_game_amd.mfg_ac2 import actor_critic
    ops().clear_status()
    rs = np.random.RandomState(0)
    ac = actor_critic(theta=8.86349, d=21, pi0=rs.dirichlet(np.ones(21), size=4), batch=32, seed=1, update_every='rollout',
                      verbose=0)
    ac.theta = 500.0
    with pytest.raises(L.MfgError, match='mfg_clear_status'):
        ac.train(num_episodes=3)
    assert ops().status() == 0                  # the condition sits in the INSTANCE's word, not in the caller's default context
    ac.clear_status()
    ac.theta = 8.86349
    ac.w = np.zeros_like(ac.w)
    ac.train(num_episodes=1)
    assert np.isfinite(np.ravel(ac.theta)[0])
