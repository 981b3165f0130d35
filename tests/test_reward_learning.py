"""CPU checks of the reward-learning infrastructure: the fp64 analytic gradient added to
oracle/reward_net_oracle.py against PyTorch autograd of networks.RewardNet and central differences, the oracle's
tf.train.AdamOptimizer formula, the device store's FIFO bookkeeping (on CPU tensors), and the flat parameter layout the
HIP training step expects (host-only entry points; no compute without a GPU)."""
import ctypes as C

import numpy as np
import pytest
import torch

from discrete_mean_field_game_amd.networks import REG_VARIANTS, RewardNet, maxent_irl_loss
from discrete_mean_field_game_amd.reward_learning import TrajectoryStore
from oracle import reward_net_oracle as RO


def _batch(rs, d, n):
    return rs.dirichlet(np.ones(d), size=n), rs.dirichlet(np.ones(d) * 0.5, size=(n, d))


@pytest.mark.parametrize('reg', ['none', 'l1l2'])
@pytest.mark.parametrize('d,k1,f2,k2', [(7, 5, 2, 3), (5, 3, 1, 5)])
def test_oracle_gradient_equals_autograd(reg, d, k1, f2, k2):
    torch.manual_seed(0)
    net = RewardNet(d=d, reg=reg, k1=k1, f2=f2, k2=k2, n_fc3=5, n_fc4=3).double()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.05 * torch.randn_like(p))
    rs = np.random.RandomState(1)
    nd, ng, T = 2, 3, 15
    ds, da = _batch(rs, d, nd * T)
    gs, ga = _batch(rs, d, ng * T)
    (loss, first, second, reg_v), g, _ = RO.irl_loss_and_grad(RO.params_from_torch(net), ds, da, gs, ga, 5, ng, l1l2=net.use_l1l2)
    tl, tf_, ts = maxent_irl_loss(net(torch.tensor(ds), torch.tensor(da)), net(torch.tensor(gs), torch.tensor(ga)), 5, ng,
                                  net.regularization() if net.use_l1l2 else None)
    grads = torch.autograd.grad(tl, list(net.parameters()))
    ref = np.concatenate([x.numpy().reshape(-1) for x in grads])
    got = RO.flatten_like_kernel(g)
    assert abs(float(tl) - loss) < 1e-12 and abs(float(tf_) - first) < 1e-12 and abs(float(ts) - second) < 1e-12
    assert np.max(np.abs(got - ref)) <= 1e-12 * max(1.0, np.abs(ref).max())


def test_oracle_gradient_with_masks_equals_central_differences():
    """Dropout masks are constants of the graph: the analytic gradient with masks against finite differences of the masked
    forward (a few entries of every tensor)."""
    d, n3, n4 = 6, 4, 3
    torch.manual_seed(2)
    net = RewardNet(d=d, reg='dropout', n_fc3=n3, n_fc4=n4).double()
    with torch.no_grad():
        for p in net.parameters():
            p.add_(0.1 * torch.randn_like(p))
    rs = np.random.RandomState(5)
    ds, da = _batch(rs, d, 15)
    gs, ga = _batch(rs, d, 30)
    keep = 0.4
    m3 = np.where(rs.rand(45, n3) <= keep, 1 / keep, 0.0)
    m4 = np.where(rs.rand(45, n4) <= keep, 1 / keep, 0.0)
    prm = RO.params_from_torch(net)
    _, g, _ = RO.irl_loss_and_grad(prm, ds, da, gs, ga, 5, 2, masks=(m3, m4))

    def loss_of(pp):
        return RO.irl_loss_and_grad(pp, ds, da, gs, ga, 5, 2, masks=(m3, m4))[0][0]
    for name in RO.FLAT_ORDER:
        flat = prm[name].reshape(-1)
        for k in rs.choice(flat.size, size=min(4, flat.size), replace=False):
            pp = {n: v.copy() for n, v in prm.items()}
            h = 1e-6
            pp[name].reshape(-1)[k] += h
            up = loss_of(pp)
            pp[name].reshape(-1)[k] -= 2 * h
            dn = loss_of(pp)
            fd = (up - dn) / (2 * h)
            assert abs(fd - g[name].reshape(-1)[k]) <= 1e-6 * max(1.0, abs(fd)), (name, k)


def test_oracle_adam_is_tf_formula():
    """One hand-computed step of tf.train.AdamOptimizer (epsilon outside the bias correction: 'epsilon hat')."""
    p, g = np.array([1.0, -2.0]), np.array([0.5, -1e-9])
    p1, m1, v1 = RO.adam_tf(p, g, np.zeros(2), np.zeros(2), 1, lr=1e-2)
    lr_t = 1e-2 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.allclose(m1, 0.1 * g) and np.allclose(v1, 0.001 * g * g)
    assert np.allclose(p1, p - lr_t * m1 / (np.sqrt(v1) + 1e-8), rtol=0, atol=1e-18)
    assert abs((p - p1)[0] - 1e-2) < 1e-8                    # |g| >> eps: a full lr step


def test_store_fifo_matches_list_semantics():
    d, T = 3, 15
    rs = np.random.RandomState(0)

    def mk(n):
        return [[(rs.rand(d), rs.rand(d, d)) for _ in range(T)] for _ in range(n)]
    st = TrajectoryStore(d, T, 'cpu')
    ref = mk(4)
    st.assign_list(ref)
    for it in range(25):
        n = rs.randint(0, 5)
        drop = rs.randint(0, len(ref) + n + 1) if it % 3 == 0 else min(n, len(ref))
        new = mk(n)
        s = torch.tensor(np.array([[p[0] for p in t] for t in new]).reshape(n, T, d), dtype=torch.float32)
        a = torch.tensor(np.array([[p[1] for p in t] for t in new]).reshape(n, T, d, d), dtype=torch.float32)
        v0 = st.version
        st.push(s, a, drop=drop)
        assert st.version == v0 + 1
        ref = (ref + new)[drop:]                              # ac_irl.py:929-932
        got = st.to_list()
        assert got is st.to_list() and len(got) == len(ref)   # cached view
        for x, y in zip(got, ref):
            for (p0, P0), (p1, P1) in zip(x, y):
                assert np.array_equal(p0, p1.astype(np.float32)) and np.array_equal(P0, P1.astype(np.float32))
        assert len(set(st.rows)) == len(st.rows) and not (set(st.rows) & set(st._free))
    with pytest.raises(ValueError):
        st.assign_list([[(np.zeros(d), np.zeros((d, d)))] * 3])


def test_flat_layout_of_the_training_step_is_the_module_parameter_order():
    from discrete_mean_field_game_amd import _lib as L
    lib = L.lib()
    for d, k1, f2, k2, n3, n4 in [(21, 5, 2, 3, 8, 4), (15, 5, 2, 3, 8, 4), (9, 3, 1, 5, 6, 7)]:
        net = RewardNet(d=d, k1=k1, f2=f2, k2=k2, n_fc3=n3, n_fc4=n4)
        offs = (C.c_int64 * 11)()
        assert lib.mfg_reward_net_param_offsets(d, k1, f2, k2, n3, n4, offs) == 0
        sizes = [p.numel() for p in net.parameters()]
        assert list(offs) == list(np.cumsum([0] + sizes))
        assert lib.mfg_reward_net_num_params(d, k1, f2, k2, n3, n4) == sum(sizes)
        N = 150
        need = lib.mfg_reward_net_train_workspace_bytes(d, k1, f2, k2, n3, n4, N)
        assert need >= 4 * N * (1 + f2 * d * d + n3 + sum(sizes) - n3 * f2 * d * d)
    assert sum(p.numel() for p in RewardNet(d=21).parameters()) == 7235     # "~7k floats"
