"""Host-side reward network (PyTorch) vs its NumPy restatement + hand-derived values (CPU).
The reference's TF net is "parity unpinned" (see oracle/reward_net_oracle.py)."""
import numpy as np
import pytest
import torch

from discrete_mean_field_game_amd.networks import RewardNet, maxent_irl_loss, REG_VARIANTS
from oracle import reward_net_oracle as RO


@pytest.mark.parametrize('reg', REG_VARIANTS)
@pytest.mark.parametrize('d', [15, 21])
def test_forward_matches_numpy_restatement(reg, d):
    torch.manual_seed(0)
    net = RewardNet(d=d, reg=reg, dropout_always=False).double().eval()
    for p in net.parameters():                                   # non-zero biases exercise every term
        if p.dim() == 1:
            torch.nn.init.uniform_(p, -0.1, 0.1)
    rs = np.random.RandomState(1)
    state = rs.dirichlet(np.ones(d), size=7)
    action = rs.dirichlet(np.ones(d), size=(7, d))
    out = net(torch.as_tensor(state), torch.as_tensor(action)).detach().numpy()
    ref = RO.forward(RO.params_from_torch(net), state, action)
    assert out.shape == (7, 1)
    assert np.allclose(out, ref, rtol=1e-10, atol=1e-12)
    assert np.all(np.abs(out) < 1)                               # tanh range
    reg_t = float(net.regularization())
    assert np.isclose(reg_t, RO.l1_l2(RO.params_from_torch(net)) if 'l1l2' in reg else 0.0)


def test_parameter_count_and_init():
    net = RewardNet(d=21, n_fc3=8, n_fc4=4)
    assert sum(p.numel() for p in net.parameters()) == 26 + 20 + 7064 + 120 + 5
    assert all(float(m.bias.abs().sum()) == 0 for m in (net.conv1, net.conv2, net.fc3, net.fc4, net.out))
    lim = np.sqrt(6.0 / (2 * 21 * 21 + 8))
    assert float(net.fc3.weight.abs().max()) <= lim + 1e-7       # Xavier uniform


def test_dropout_active_at_eval_like_reference():
    torch.manual_seed(3)
    net = RewardNet(d=15, reg='dropout_l1l2').eval()             # dropout_always=True is the default
    s = torch.rand(64, 15); a = torch.rand(64, 15, 15)
    assert not torch.equal(net(s, a), net(s, a))
    net2 = RewardNet(d=15, reg='dropout_l1l2', dropout_always=False).eval()
    assert torch.equal(net2(s, a), net2(s, a))


def test_maxent_loss_hand_values():
    # constant rewards: first = -(1/5)*75*c, second = log(mean exp(15 c)) = 15 c
    c = 0.2
    r_demo = torch.full((75, 1), c); r_gen = torch.full((75, 1), c)
    loss, first, second = maxent_irl_loss(r_demo, r_gen, 5, 5)
    assert np.isclose(float(first), -15 * c) and np.isclose(float(second), 15 * c) and abs(float(loss)) < 1e-6
    rs = np.random.RandomState(0)
    rd, rg = rs.uniform(-1, 1, (75, 1)), rs.uniform(-1, 1, (75, 1))
    loss, first, second = maxent_irl_loss(torch.as_tensor(rd), torch.as_tensor(rg), 5, 5, torch.tensor(0.5))
    ref = RO.irl_loss(rd, rg, 5, 5, 0.5)
    assert np.allclose([float(loss), float(first), float(second)], ref, rtol=1e-12)


def test_gemm_conv_equals_conv2d():
    from discrete_mean_field_game_amd.networks import conv2d_same_gemm
    torch.manual_seed(0)
    x = torch.randn(5, 2, 21, 21, dtype=torch.float64, requires_grad=True)
    w = torch.randn(3, 2, 5, 5, dtype=torch.float64, requires_grad=True)
    b = torch.randn(3, dtype=torch.float64)
    a = conv2d_same_gemm(x, w, b)
    r = torch.nn.functional.conv2d(x, w, b, padding=2)
    assert torch.allclose(a, r, rtol=1e-12, atol=1e-12)
    ga = torch.autograd.grad(a.sum(), (x, w))
    gr = torch.autograd.grad(torch.nn.functional.conv2d(x, w, b, padding=2).sum(), (x, w))
    assert all(torch.allclose(p, q, rtol=1e-10, atol=1e-10) for p, q in zip(ga, gr))


def test_oracle_conv_is_tf_same_cross_correlation():
    """f1 hardening: the two semantic risks of the TF restatement checked against THIRD-PARTY code (scipy), with
    asymmetric kernels so that a flipped kernel or a transposed layout cannot pass: (a) conv2d SAME, stride 1 is a
    cross-correlation with (k-1)/2 zeros on each side == scipy.signal.correlate2d(mode='same', boundary='fill');
    (b) the NHWC flatten feeding fc3 (networks.py:67) orders features as (h, w, channel)."""
    from scipy.signal import correlate2d
    rs = np.random.RandomState(5)
    N, H, Cin, Cout = 3, 21, 2, 3
    for k in (3, 5):
        x = rs.randn(N, H, H, Cin)
        w = rs.randn(k, k, Cin, Cout)                             # TF layout HWIO, no symmetry
        b = rs.randn(Cout)
        got = RO.conv2d_same(x, w, b)
        for n in range(N):
            for o in range(Cout):
                ref = sum(correlate2d(x[n, :, :, c], w[:, :, c, o], mode='same', boundary='fill', fillvalue=0.0)
                          for c in range(Cin)) + b[o]
                assert np.allclose(got[n, :, :, o], ref, rtol=1e-12, atol=1e-12)
    # (b) flatten order: put a single 1 at (h, w, c) of the conv2 output position by construction -- a net whose convs are
    # identities on channel 0 / zero on channel 1 -- and read which fc3 input column it reaches
    d = 6
    params = {'conv1_w': np.zeros((5, 5, 1, 1)), 'conv1_b': np.zeros(1), 'conv2_w': np.zeros((3, 3, 1, 2)),
              'conv2_b': np.zeros(2), 'fc3_w': np.zeros((2 * d * d, 1)), 'fc3_b': np.zeros(1),
              'fc4_w': np.zeros((1 + d, 1)), 'fc4_b': np.zeros(1), 'out_w': np.ones((1, 1)), 'out_b': np.zeros(1)}
    params['conv1_w'][2, 2, 0, 0] = 1.0                            # identity
    params['conv2_w'][1, 1, 0, 1] = 1.0                            # channel 1 = identity, channel 0 = 0
    params['fc4_w'][0, 0] = 1.0
    h, w_ = 2, 4
    action = np.zeros((1, d, d)); action[0, h, w_] = 0.7
    for col in range(2 * d * d):
        params['fc3_w'][:] = 0.0
        params['fc3_w'][col, 0] = 1.0
        out = float(RO.forward(params, np.zeros((1, d)), action)[0, 0])
        want = np.tanh(0.7) if col == (h * d + w_) * 2 + 1 else 0.0
        assert abs(out - want) < 1e-15, (col, out)


@pytest.mark.parametrize('reg', ['none', 'dropout_l1l2'])
def test_tf_checkpoint_layout_round_trip(reg):
    """RewardNet.tf_variables / load_tf_variables: HWIO <-> OIHW, [in,out] <-> [out,in], names of the reference's scopes
    (networks.py:62-79 under variable_scope 'reward', ac_irl.py:246).  A net loaded from TF-layout arrays computes what the
    TF-layout oracle computes FROM THOSE ARRAYS -- the converter is not checked against itself."""
    d, n3, n4 = 15, 8, 4
    rs = np.random.RandomState(2)
    tfv = {'reward/conv1/weights': rs.randn(5, 5, 1, 1), 'reward/conv1/biases': rs.randn(1) * 0.1,
           'reward/conv2/weights': rs.randn(3, 3, 1, 2), 'reward/conv2/biases': rs.randn(2) * 0.1,
           'reward/fc3/weights': rs.randn(2 * d * d, n3) * 0.05, 'reward/fc3/biases': rs.randn(n3) * 0.1,
           'reward/fc4/weights': rs.randn(n3 + d, n4) * 0.3, 'reward/fc4/biases': rs.randn(n4) * 0.1,
           'reward/out/weights': rs.randn(n4, 1), 'reward/out/biases': rs.randn(1) * 0.1}
    tfv = {k: v.astype(np.float32) for k, v in tfv.items()}
    net = RewardNet(d=d, reg=reg, n_fc3=n3, n_fc4=n4, dropout_always=False).eval().load_tf_variables(tfv)
    back = net.tf_variables()
    assert sorted(back) == sorted(tfv)
    for k in tfv:
        assert back[k].shape == tfv[k].shape and np.array_equal(back[k], tfv[k])
    state = rs.dirichlet(np.ones(d), size=5)
    action = rs.dirichlet(np.ones(d), size=(5, d))
    params = {k.split('/')[1] + ('_w' if k.endswith('weights') else '_b'): v.astype(np.float64) for k, v in tfv.items()}
    ref = RO.forward(params, state, action)
    out = net.double()(torch.as_tensor(state), torch.as_tensor(action)).detach().numpy()
    assert np.allclose(out, ref, rtol=1e-10, atol=1e-12)
    with pytest.raises(ValueError):
        RewardNet(d=d + 1, reg=reg, n_fc3=n3, n_fc4=n4).load_tf_variables(tfv)
