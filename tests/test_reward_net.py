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
