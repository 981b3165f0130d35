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
