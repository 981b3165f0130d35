"""-m gpu tests at BASELINE.json's FULL sizes (target point d=21/T=15/B=65536, C3 d=128/B=16384, C5's d=256).

The oracle cannot run these sizes in seconds, so the checks are the size-independent properties the domain offers:
row-stochastic actions, mass conservation of pi' = P^T pi, linearity of the transition in pi, agreement of the fused
rollout with the unfused given-P kernel on its own materialised actions, invariance of the Philox stream to how the
batch is split across launches / ranks (`traj_offset`), a checksum of checksums for the batch gradient (fp64 torch
recomputation of G from the per-sample outputs), and the oracle on a random subsample of trajectories.
Tolerances: bit exact where both sides are the kernels' own fp64-accumulate/round-once arithmetic; 1e-5 relative for
rewards / TD errors against the oracle (BASELINE.json north star); 1e-9 for fp64 sums of identical terms.
"""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

pytestmark = pytest.mark.gpu

THETA, SHIFT, SCALE = 8.86349, 0.16, 12000.0       # mfg_ac2.py:832


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


def start_states(B, d, dev, seed=0):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    x = -torch.log(torch.rand(B, d, device=dev, generator=g).clamp_min(1e-12))     # Dirichlet(1) rows
    return (x / x.sum(1, keepdim=True)).float().contiguous()


def full_rollout(dev, d, B, T, seed=11, **kw):
    o = ops()
    pi0 = start_states(B, d, dev)
    theta = torch.tensor([THETA], dtype=torch.float64, device=dev)
    w = torch.as_tensor(np.random.RandomState(1).rand(o.num_features(d)), device=dev)
    out = o.rollout(pi0, T, theta, SHIFT, SCALE, w=w, gamma=0.9, seed=seed, td=True, write_P=True, **kw)
    torch.cuda.synchronize()
    return pi0, theta, w, out


@pytest.mark.parametrize('d,B,T', [(21, 65536, 15), (128, 16384, 2), (256, 2048, 2)])
def test_fullsize_rollout_properties(dev, d, B, T):
    o = ops()
    pi0, theta, w, out = full_rollout(dev, d, B, T)
    P, traj, r, delta, g, G = out['P'], out['pi_traj'], out['reward'], out['delta'], out['g'], out['G']
    assert torch.isfinite(P).all() and torch.isfinite(traj).all() and torch.isfinite(delta).all() and torch.isfinite(g).all()
    # actions are row-stochastic, non-negative
    assert float(P.min()) >= 0.0
    rows = P.double().sum(-1)
    assert float((rows - 1.0).abs().max()) < 4e-6
    # mass conservation (pi' = P^T pi with rows of P summing to 1)
    mass = traj.double().sum(-1)
    assert float((mass - mass[:, :1]).abs().max()) < 1e-5
    assert torch.equal(traj[:, 0], pi0) and torch.equal(traj[:, T], out['pi_last'])
    # fused == unfused: the given-P kernel on the materialised actions reproduces pi' bit for bit, r to 1e-6
    N = B * T
    pn, rr = o.step_given_P(traj[:, :T].contiguous().view(N, d), P.view(N, d, d))
    assert torch.equal(pn.view(B, T, d), traj[:, 1:])
    if d in (21, 15):
        # the packed compile-time-d kernels share the column-pass arithmetic and summation order: rewards bit for bit
        assert torch.equal(rr.view(B, T), r)
    else:
        assert float(((rr.view(B, T) - r).abs() / r.abs().clamp_min(1e-3)).max()) < 2e-6
    # checksum of checksums: the batch gradient equals an fp64 recomputation from the per-sample outputs
    x = traj[:, :T].double().reshape(N, d)
    dl = delta.reshape(N)
    M = (x * dl[:, None]).T @ x
    iu = torch.triu_indices(d, d, device=dev)
    Q = d * (d + 1) // 2
    Gq = G[:Q]
    ref_q = M[iu[0], iu[1]]                                   # row-major upper triangle == feature order (mfg_ac2.py:325-344)
    scale = float(ref_q.abs().max())
    assert float((Gq - ref_q).abs().max()) <= 1e-9 * scale
    assert float((G[Q:Q + d] - (x * dl[:, None]).sum(0)).abs().max()) <= 1e-9 * max(1.0, float(dl.abs().sum()))
    assert abs(float(G[Q + d]) - float(dl.sum())) <= 1e-9 * float(dl.abs().sum())
    assert abs(float(G[Q + d + 1]) - float((dl * g.reshape(N)).sum())) <= 1e-9 * float((dl * g.reshape(N)).abs().sum())
    assert abs(float(G[Q + d + 2]) - float(r.double().sum())) <= 1e-9 * float(r.double().abs().sum())
    assert float(G[Q + d + 3]) == float(N)


def test_fullsize_subsample_against_oracle(dev):
    """64 random trajectories of the B=65536 rollout, every step, against the NumPy oracle on the same actions."""
    d, B, T = 21, 65536, 15
    pi0, theta, w, out = full_rollout(dev, d, B, T)
    sel = torch.as_tensor(np.random.RandomState(5).choice(B, 64, replace=False), device=dev)
    P = out['P'][sel].cpu().numpy().astype(np.float64)
    traj, r, dl, gg, _, _ = O().batched_rollout_given_P(pi0[sel].cpu().numpy(), P, w.cpu().numpy(), THETA, SHIFT, gamma=0.9)
    np.testing.assert_allclose(out['pi_traj'][sel].cpu().numpy(), traj, rtol=0, atol=1e-7)
    got_r = out['reward'][sel].cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got_r - r) / np.maximum(np.abs(r), 1e-3)) < 1e-5
    got_d = out['delta'][sel].cpu().numpy()
    assert np.max(np.abs(got_d - dl) / np.maximum(np.abs(dl), 1e-2)) < 1e-5
    got_g = out['g'][sel].cpu().numpy()
    assert np.max(np.abs(got_g - gg) / np.maximum(np.abs(gg), 1.0)) < 1e-5


def test_fullsize_split_invariance(dev):
    """The Philox stream is keyed by the global trajectory id: a rollout of the whole batch equals rollouts of its
    shards with traj_offset (what each rank of an N-GPU job runs), bit for bit, and the shard gradients add up."""
    o = ops()
    d, B, T = 21, 65536, 15
    pi0, theta, w, whole = full_rollout(dev, d, B, T)
    cuts = [0, 8192, 8192 + 12345, B]                              # ragged on purpose (not multiples of the tile)
    Gsum = torch.zeros_like(whole['G'])
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = o.rollout(pi0[lo:hi].contiguous(), T, theta, SHIFT, SCALE, w=w, gamma=0.9, seed=11, traj_offset=lo,
                         td=True, write_P=False)
        assert torch.equal(part['pi_traj'], whole['pi_traj'][lo:hi])
        assert torch.equal(part['reward'], whole['reward'][lo:hi])
        assert torch.equal(part['delta'], whole['delta'][lo:hi])
        assert torch.equal(part['g'], whole['g'][lo:hi])
        Gsum += part['G']
    scale = whole['G'].abs().clamp_min(1e-6)
    assert float(((Gsum - whole['G']).abs() / scale).max()) < 1e-9
    # WRITE_P does not change the trajectory
    assert torch.equal(whole['pi_traj'], o.rollout(pi0, T, theta, SHIFT, SCALE, seed=11, td=False)['pi_traj'])


@pytest.mark.parametrize('d,B', [(21, 983040), (128, 16384), (256, 4096)])
def test_fullsize_transition_linearity_and_reward_kinds(dev, d, B):
    """step_given_P at the bench's slab sizes: linear in pi, synthetic reward = -1/2 sum_i pi_i |P_i|^2, and the
    reward-less call returns the same pi'."""
    o = ops()
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    P = torch.rand(B, d, d, device=dev, generator=g)
    P /= P.sum(-1, keepdim=True)
    a, b = start_states(B, d, dev, 1), start_states(B, d, dev, 2)
    pa, ra = o.step_given_P(a, P)
    pb, _ = o.step_given_P(b, P)
    mix = (0.25 * a + 0.75 * b).contiguous()
    pm, _ = o.step_given_P(mix, P)
    assert float((pm - (0.25 * pa + 0.75 * pb)).abs().max()) < 3e-7
    assert float((pm.double().sum(-1) - mix.double().sum(-1)).abs().max()) < 2e-6
    from discrete_mean_field_game_amd import _lib as L
    p2, rs = o.step_given_P(a, P, reward_kind=L.REWARD_SYNTHETIC)
    assert torch.equal(p2, pa)
    idx = torch.arange(0, B, max(1, B // 4096), device=dev)       # fp64 torch check on a strided subsample
    ref_s = -0.5 * (a[idx].double() * (P[idx].double() ** 2).sum(-1)).sum(-1)
    assert float(((rs[idx].double() - ref_s).abs() / ref_s.abs().clamp_min(1e-6)).max()) < 1e-6
    Pd, ad = P[idx].double(), a[idx].double()
    ref_r = (ad[:, :, None] * Pd ** 2 * (ad[:, None, :] - ad[:, :, None])).sum((1, 2))   # mfg_ac2.py:257-287
    assert float(((ra[idx].double() - ref_r).abs() / ref_r.abs().clamp_min(1e-4)).max()) < 1e-5
    p3, none = o.step_given_P(a, P, want_reward=False)
    assert none is None and torch.equal(p3, pa)


# ---- BASELINE.json configs C3 / C5 (per-GPU share) / C4 at their stated (d, T, B) ----------------------------------------
def _config_rollout(o, pi0, theta, w, T, seed, lo=0, hi=None, write_P=False):
    hi = pi0.shape[0] if hi is None else hi
    out = o.rollout(pi0[lo:hi].contiguous(), T, theta, SHIFT, SCALE, w=w, gamma=0.9, seed=seed, traj_offset=lo, td=True,
                    write_P=write_P)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize('d,B,T,nsub', [(128, 16384, 40, 32), (256, 16384, 40, 16)], ids=['C3', 'C5_share'])
def test_config_full_T_large_d(dev, d, B, T, nsub):
    """C3 (d=128, T=40, B=16384) and C5's per-GPU share (d=256, T=40, B=131072/8) at their FULL (d, T, B), fused TD
    rollout without materialised actions (4.3 GB / step at d=256): finiteness, mass conservation over all 40 steps,
    run-to-run determinism, the gradient checksum, invariance to a ragged split of the batch (what ranks run), and --
    because the Philox stream is keyed by (global trajectory id, step) -- an `nsub`-trajectory slice re-run with
    traj_offset + WRITE_P reproduces the big run bit for bit and is then replayed step by step by the oracle
    (pi_traj indexing, running state, Philox step counters at T=40 on the large-d kernel)."""
    o = ops()
    pi0 = start_states(B, d, dev)
    theta = torch.tensor([THETA], dtype=torch.float64, device=dev)
    w = torch.as_tensor(np.random.RandomState(1).rand(o.num_features(d)), device=dev)
    whole = _config_rollout(o, pi0, theta, w, T, seed=23)
    traj, r, delta, g, G = whole['pi_traj'], whole['reward'], whole['delta'], whole['g'], whole['G']
    assert tuple(traj.shape) == (B, T + 1, d) and tuple(r.shape) == (B, T)
    assert torch.isfinite(traj).all() and torch.isfinite(r).all() and torch.isfinite(delta).all() and torch.isfinite(g).all()
    assert float(traj.min()) >= 0.0
    mass = traj.double().sum(-1)
    assert float((mass - mass[:, :1]).abs().max()) < 2e-5
    assert torch.equal(traj[:, 0], pi0) and torch.equal(traj[:, T], whole['pi_last'])
    # determinism
    again = _config_rollout(o, pi0, theta, w, T, seed=23)
    for k in ('pi_traj', 'reward', 'delta', 'g', 'G'):
        assert torch.equal(whole[k], again[k]), k
    del again
    # gradient checksum (fp64 recomputation of the sums from the per-sample outputs)
    N = B * T
    x = traj[:, :T].double().reshape(N, d)
    dl = delta.reshape(N)
    M = (x * dl[:, None]).T @ x
    iu = torch.triu_indices(d, d, device=dev)
    Q = d * (d + 1) // 2
    ref_q = M[iu[0], iu[1]]
    assert float((G[:Q] - ref_q).abs().max()) <= 1e-9 * float(ref_q.abs().max())
    assert float((G[Q:Q + d] - (x * dl[:, None]).sum(0)).abs().max()) <= 1e-9 * max(1.0, float(dl.abs().sum()))
    assert abs(float(G[Q + d + 1]) - float((dl * g.reshape(N)).sum())) <= 1e-9 * float((dl * g.reshape(N)).abs().sum())
    assert abs(float(G[Q + d + 2]) - float(r.double().sum())) <= 1e-9 * float(r.double().abs().sum())
    assert float(G[Q + d + 3]) == float(N)
    del x, M
    # ragged split: shards with traj_offset reproduce the whole run; their gradient buffers add up
    cuts = [0, 5000, 5000 + 4097, B]
    Gsum = torch.zeros_like(G)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = _config_rollout(o, pi0, theta, w, T, seed=23, lo=lo, hi=hi)
        assert torch.equal(part['pi_traj'], traj[lo:hi]) and torch.equal(part['delta'], delta[lo:hi])
        assert torch.equal(part['g'], g[lo:hi]) and torch.equal(part['reward'], r[lo:hi])
        Gsum += part['G']
        del part
    assert float(((Gsum - G).abs() / G.abs().clamp_min(1e-6)).max()) < 1e-9
    # slice with materialised actions == the big run; oracle replay of the slice on its own actions
    lo = 7777
    sub = _config_rollout(o, pi0, theta, w, T, seed=23, lo=lo, hi=lo + nsub, write_P=True)
    for k in ('pi_traj', 'reward', 'delta', 'g'):
        assert torch.equal(sub[k], whole[k][lo:lo + nsub]), k
    P = sub['P'].cpu().numpy()
    rows = P.astype(np.float64).sum(-1)
    assert P.min() >= 0.0 and np.abs(rows - 1.0).max() < 4e-6
    otraj, orr, odl, ogg, _, _ = O().batched_rollout_given_P(pi0[lo:lo + nsub].cpu().numpy(), P, w.cpu().numpy(), THETA,
                                                            SHIFT, gamma=0.9)
    np.testing.assert_allclose(sub['pi_traj'].cpu().numpy(), otraj, rtol=0, atol=1e-7)
    got_r = sub['reward'].cpu().numpy().astype(np.float64)
    assert np.max(np.abs(got_r - orr) / np.maximum(np.abs(orr), 1e-4)) < 1e-5
    got_d = sub['delta'].cpu().numpy()
    assert np.max(np.abs(got_d - odl) / np.maximum(np.abs(odl), 1e-2)) < 1e-5
    got_g = sub['g'].cpu().numpy()
    assert np.max(np.abs(got_g - ogg) / np.maximum(np.abs(ogg), 1.0)) < 1e-5


def _forward_with_masks(RO, params, pi, P, m3, m4):
    """reward_net_oracle.forward with explicit dropout masks (rows picked out of a larger call's mask block)."""
    N, d = pi.shape
    x = P.reshape(N, d, d, 1).astype(np.float64)
    x = np.maximum(RO.conv2d_same(x, params['conv1_w'], params['conv1_b']), 0)
    x = np.maximum(RO.conv2d_same(x, params['conv2_w'], params['conv2_b']), 0)
    x = np.maximum(x.reshape(N, -1).dot(params['fc3_w']) + params['fc3_b'], 0) * m3
    x = np.concatenate([x, pi.astype(np.float64)], axis=1)
    x = np.maximum(x.dot(params['fc4_w']) + params['fc4_b'], 0) * m4
    return np.tanh(x.dot(params['out_w']) + params['out_b'])[:, 0]


@pytest.mark.parametrize('precision,tol,reg', [('f64', 5e-8, 'l1l2'), ('mixed', 2e-6, 'l1l2'), ('mixed', 2e-6, 'dropout_l1l2')])
@pytest.mark.parametrize('mode', ['step', 'rollout'])
def test_config_C4_irl_train_with_reward_net_in_the_loop(dev, mode, precision, tol, reg):
    """C4: AC_IRL.train at d=21, B=4096 with the HIP reward-net kernel inside the loop (reg='l1l2': no dropout, so the
    run is replayable), one episode with gamma=0.9, replayed by the oracle: actions re-drawn from the same Philox
    counters, reward = oracle/reward_net_oracle.forward on them, 1-indexed episode schedule, running discount
    (ac_irl.py:664-712), batch-mean updates per step / once per episode.  Step mode runs through the native episode loop
    (mfg_train_episode_irl).  precision 'mixed' (the default of the class): the oracle's score is exact fp64, the kernel's
    is within ~2e-7 of it, so the replayed parameters agree a little less tightly.  reg='dropout_l1l2' is the class
    default (ac_irl.py:33): dropout stays on when the net serves as the RL reward, and the oracle redraws the kernel's
    counter-based masks (reward_net_oracle.dropout_masks) with the keys the host class documents -- the default path is
    replayed end to end, not only checked statistically."""
    from discrete_mean_field_game_amd.ac_irl import AC_IRL
    from oracle import reward_net_oracle as RO
    o = ops()
    d, B, gamma = 21, 4096, 0.9
    rs = np.random.RandomState(8)
    mat = rs.dirichlet(np.ones(d), size=64)
    np.random.seed(31); torch.manual_seed(31)
    ac = AC_IRL(theta=8.64, shift=0.0, alpha_scale=1e4, d=d, pi0=mat, demonstrations=[], batch=B, rng='philox', seed=13,
                reg=reg, update_every=mode, precision=precision, verbose=0)
    with torch.no_grad():                                           # non-trivial biases (zero by default)
        for p in ac.reward_net.parameters():
            if p.dim() == 1:
                p.uniform_(-0.2, 0.2)
    assert o.reward_net_supported(ac.reward_net)
    params = RO.params_from_torch(ac.reward_net)
    w0 = ac.w[:, 0].copy(); theta0 = float(np.ravel(ac.theta)[0])
    np.random.seed(32)
    ac.train(max_episodes=1, stop_criteria=-1, gamma=gamma, constant=False, lr_critic=0.1, lr_actor=0.001)
    assert ac._reward_calls == (15 if mode == 'step' else 1)        # the kernel path ran (not the torch module)
    from oracle.philox_ref import start_indices
    idx = start_indices(13, 0, np.arange(B), 64)                    # the device-side start draw of the episode
    pi = mat[idx].astype(np.float32)
    w, theta = w0.copy(), theta0
    sc, sa = O().lr_scales(1, False)
    Gw_acc = np.zeros_like(w); Gt_acc = 0.0
    disc = 1.0
    for t in range(15):
        th = torch.tensor([theta], dtype=torch.float64, device=dev)
        P = o.sample_dirichlet(torch.as_tensor(pi, device=dev), th, 0.0, 1e4, seed=13, step=t, precision=precision).cpu().numpy()
        pn = O().transition(P, pi).astype(np.float32)
        drop = None
        if 'dropout' in reg and mode == 'step':
            # AC_IRL.reward(): Philox key = (seed + 0x5EED) ^ (call number * 0x9E3779B97F4A7C15), sample counter = trajectory
            drop = (0.4, ((13 + 0x5EED) ^ ((t + 1) * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF, 0)
        r = RO.forward(params, pi.astype(np.float64), P.astype(np.float64), dropout=drop)[:, 0]
        if 'dropout' in reg and mode == 'rollout':
            # one reward() call over all B*T transitions of the episode, sample n = b*T + t: redraw its masks for step t
            key = ((13 + 0x5EED) ^ (1 * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
            m3, m4 = RO.dropout_masks(0.4, key, 0, B * 15, 8, 4)
            r = _forward_with_masks(RO, params, pi, P, m3[t::15], m4[t::15])
        r = r.astype(np.float32).astype(np.float64)                 # the kernel hands the reward over as fp32
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
    # fp32 reward kernel vs fp64 restatement: rewards agree to ~1e-6 absolute, so the updates agree to ~1e-9
    assert abs(float(np.ravel(ac.theta)[0]) - theta) < tol
    assert np.max(np.abs(ac.w[:, 0] - w)) < tol


@pytest.mark.parametrize('d,B,T', [(80, 7, 3), (128, 5, 6), (144, 33, 2), (256, 3, 5), (512, 2, 2), (128, 700, 2), (192, 301, 1),
                                   (256, 131, 3)])
def test_matrix_core_values_equal_in_kernel_values(dev, d, B, T):
    """Large-d TD rollouts evaluate V of all B (T+1) states on the fp64 matrix cores after the rollout (k_value_mfma;
    k_value_mfma2 when d is a multiple of 64: 128 states per block -- the larger cases span several passes per block) when
    a workspace is passed; without one (the IRL form, reward_kind EXTERNAL) the rollout kernel evaluates the same values
    itself.  Both must give the same TD errors (re-associated fp64 sums: 1e-12), for both discount forms and ragged
    state counts (B (T+1) not a multiple of the 16-state tile), and match the oracle."""
    from discrete_mean_field_game_amd import _lib as L
    o = ops()
    pi0 = start_states(B, d, dev, seed=4)
    theta = torch.tensor([THETA], dtype=torch.float64, device=dev)
    w = torch.as_tensor(np.random.RandomState(2).rand(o.num_features(d)), device=dev)
    for dpow in (False, True):
        a = o.rollout(pi0, T, theta, SHIFT, SCALE, w=w, gamma=0.9, seed=3, td=True, discount_pow=dpow)           # deferred
        b = o.rollout(pi0, T, theta, SHIFT, SCALE, w=w, gamma=0.9, seed=3, td=True, discount_pow=dpow,
                      reward_kind=L.REWARD_EXTERNAL)                                                               # in-kernel
        torch.cuda.synchronize()
        assert torch.equal(a['pi_traj'], b['pi_traj']) and torch.equal(a['g'], b['g'])
        want = b['delta'] + a['reward'].double()
        assert float((a['delta'] - want).abs().max()) <= 1e-12 * float(want.abs().max())
    # oracle values of the stored states
    x = a['pi_traj'].cpu().numpy().astype(np.float64)
    V = O().calc_value(x.reshape(-1, d), w.cpu().numpy()).reshape(B, T + 1)
    disc = 0.9 ** np.arange(T)
    want = a['reward'].cpu().numpy().astype(np.float64) + disc[None, :] * V[:, 1:] - V[:, :-1]
    assert np.max(np.abs(a['delta'].cpu().numpy() - want)) <= 1e-11 * np.max(np.abs(V))


@pytest.mark.parametrize('d,B', [(21, 96000 + 7), (21, 192000 + 5), (21, 384000 + 50), (21, 983040), (15, 128000 + 1), (15, 512000 + 3)])
def test_given_P_batched_store_kernel_equals_per_tile_kernel(dev, d, B):
    """Large batches take k_step_wave_batched (KB consecutive tiles per wave, outputs parked in LDS and written as one
    device-scope burst per super tile, the reward line summed inside the next tile's walk; KB = 8 / 16 / 32 by batch size on
    a 256-CU device; the last B mod 12 trajectories go through the ragged-capable per-tile kernel).  It must reproduce, bit
    for bit, what the per-tile kernel gives on the same trajectories (calls on chunks small enough to stay on k_step_wave),
    for both reward kinds and the reward-less call, including the ragged last super tile; a slice is checked against the
    oracle."""
    g = torch.Generator(device=dev)
    g.manual_seed(B)
    pi = torch.rand(B, d, device=dev, generator=g)
    pi = (pi / pi.sum(1, keepdim=True)).contiguous()
    P = torch.rand(B, d, d, device=dev, generator=g)
    P /= P.sum(-1, keepdim=True)
    chunk = 40000   # < 3 super tiles per wave for every KB: per-tile kernel
    for kind in (0, 1):
        pn, r = ops().step_given_P(pi, P, reward_kind=kind)
        for s in range(0, B, chunk):
            e = min(B, s + chunk)
            pn_c, r_c = ops().step_given_P(pi[s:e], P[s:e], reward_kind=kind)
            assert torch.equal(pn[s:e], pn_c) and torch.equal(r[s:e], r_c), (kind, s)
    pn2, none = ops().step_given_P(pi, P, want_reward=False)
    assert none is None and torch.equal(pn2, pn)
    sl = slice(B - 300, B)   # the ragged tail + the last full super tiles
    Pn, pin = P[sl].cpu().numpy().astype(np.float64), pi[sl].cpu().numpy().astype(np.float64)
    assert np.array_equal(pn[sl].cpu().numpy(), O().transition(Pn, pin).astype(np.float32))
    r0 = ops().step_given_P(pi, P, reward_kind=0)[1][sl].cpu().numpy()
    ref = O().calc_reward(Pn, pin)
    assert np.max(np.abs(r0 - ref) / np.maximum(np.abs(ref), 1e-6)) < 1e-6


@pytest.mark.parametrize('d,B', [(128, 16384 - 5), (128, 32768 - 100), (128, 20000), (256, 16384 - 1)])
def test_given_P_row_kernel_super_tiles_equal_per_trajectory_stores(dev, d, B):
    """d = 128 / 256 batches whose 8-trajectory super tiles fill the resident waves' rounds to >= 90 % (on a 256-CU device:
    just under a multiple of 16 384 trajectories) take the super-tile form of k_step_rows (8 consecutive trajectories per
    wave, outputs stashed in LDS and written as one device-scope burst; ragged last super tile: direct stores); other
    sizes (20 000) stay on the per-trajectory form.  It
    must reproduce bit for bit what the per-trajectory-store form gives (calls on chunks below the threshold), for both
    reward kinds and the reward-less call; a slice is checked against the oracle."""
    g = torch.Generator(device=dev)
    g.manual_seed(B + d)
    pi = torch.rand(B, d, device=dev, generator=g)
    pi = (pi / pi.sum(1, keepdim=True)).contiguous()
    P = torch.rand(B, d, d, device=dev, generator=g)
    P /= P.sum(-1, keepdim=True)
    chunk = 4096
    for kind in (0, 1):
        pn, r = ops().step_given_P(pi, P, reward_kind=kind)
        for s in range(0, B, chunk):
            e = min(B, s + chunk)
            pn_c, r_c = ops().step_given_P(pi[s:e], P[s:e], reward_kind=kind)
            assert torch.equal(pn[s:e], pn_c) and torch.equal(r[s:e], r_c), (kind, s)
    pn2, none = ops().step_given_P(pi, P, want_reward=False)
    assert none is None and torch.equal(pn2, pn)
    sl = slice(B - 20, B)
    Pn, pin = P[sl].cpu().numpy().astype(np.float64), pi[sl].cpu().numpy().astype(np.float64)
    assert np.array_equal(pn[sl].cpu().numpy(), O().transition(Pn, pin).astype(np.float32))
    r0 = ops().step_given_P(pi, P, reward_kind=0)[1][sl].cpu().numpy()
    ref = O().calc_reward(Pn, pin)
    assert np.max(np.abs(r0 - ref) / np.maximum(np.abs(ref), 1e-6)) < 1e-6
