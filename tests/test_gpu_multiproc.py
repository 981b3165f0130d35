"""-m gpu: the CLASS-level multi-rank path (actor_critic.train / AC_IRL.train with a process group), executed for real:
two FRESH child processes share GPU 0, talk over gloo, each builds the drop-in class with the GLOBAL batch and runs
train(); the batch shards by rank, the Philox streams of the actions AND of the per-episode start-state draw are keyed by
the global trajectory id, gradients meet in one all-reduce per update.  Nothing else is exchanged: `dist.broadcast` is
replaced by a function that raises inside the children, and the ranks are given DIFFERENT host seeds on purpose.

Checked: both ranks end with bit-identical (theta, w) (replicated update, no broadcast of parameters needed), two
2-rank runs agree bit for bit (deterministic), and the result equals the single-process run of the same global batch up
to the re-association of the fp64 gradient sums across ranks (the only arithmetic that depends on the world size:
|diff| <= 1e-12 relative; bit-equality across world sizes would need exact accumulation)."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _mat(d, n=7):
    return np.random.RandomState(3).dirichlet(np.ones(d), size=n)


def _run_class(kind, mode, d, B, episodes, host_seed, world):
    """Build the class, train, return (theta, w).  Called in the parent (world == 1) and in each child."""
    sys.path.insert(0, ROOT)
    import torch
    if kind == 'ac':
        from discrete_mean_field_game_amd.mfg_ac2 import actor_critic
        np.random.seed(11)                                   # same critic initialisation on every rank
        ac = actor_critic(d=d, pi0=_mat(d), batch=B, rng='philox', seed=5, update_every=mode, precision='f64', verbose=0)
        np.random.seed(host_seed)                            # (nothing in a batched Philox run reads the host stream)
        ac.train(num_episodes=episodes, gamma=0.9, constant=0)
    else:
        from discrete_mean_field_game_amd.ac_irl import AC_IRL

        def fake_reward_dev(pi, P):
            diag = torch.diagonal(P, dim1=-2, dim2=-1).double()
            return torch.tanh(5.0 * (pi.double() * diag).sum(-1) - 0.3).float().contiguous()
        np.random.seed(11)
        ac = AC_IRL(theta=8.64, shift=0.0, alpha_scale=1e4, d=d, pi0=_mat(d), demonstrations=[], batch=B, rng='philox',
                    seed=5, update_every=mode, precision='f64', use_tf=False, verbose=0)
        np.random.seed(host_seed)
        ac.train(max_episodes=episodes, stop_criteria=-1, gamma=0.9, reward_fn=fake_reward_dev)
    return float(np.ravel(ac.theta)[0]), ac.w[:, 0].copy()


def _child(rank, world, port, kind, mode, d, B, episodes, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)                                 # both ranks on GPU 0
    dist.init_process_group('gloo', rank=rank, world_size=world)

    def _no_broadcast(*a, **k):
        raise AssertionError('train() must not broadcast: start states are drawn on the device, parameters stay replicated')
    dist.broadcast = _no_broadcast
    try:
        theta, w = _run_class(kind, mode, d, B, episodes, host_seed=100 + 17 * rank, world=world)
        np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), theta=theta, w=w)
    finally:
        dist.destroy_process_group()


def _two_ranks(kind, mode, d, B, episodes, tmp_path, tag):
    import torch.multiprocessing as mp
    out_dir = str(tmp_path / tag)
    os.makedirs(out_dir)
    ctx = mp.get_context('spawn')                            # fresh interpreters: nothing inherited from the test process
    port = _free_port()
    procs = [ctx.Process(target=_child, args=(r, 2, port, kind, mode, d, B, episodes, out_dir)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    for p in procs:
        if p.is_alive():
            p.terminate()
            pytest.fail('child rank did not finish')
        assert p.exitcode == 0
    return [np.load(os.path.join(out_dir, 'rank%d.npz' % r)) for r in range(2)]


@pytest.mark.parametrize('kind,mode,d,B', [('ac', 'rollout', 21, 50), ('ac', 'step', 21, 37), ('ac', 'rollout', 100, 9),
                                             ('irl', 'step', 15, 21), ('irl', 'rollout', 15, 21)])
def test_class_train_two_ranks_on_one_gpu(kind, mode, d, B, tmp_path):
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU')
    episodes = 2
    a = _two_ranks(kind, mode, d, B, episodes, tmp_path, 'a')
    assert float(a[0]['theta']) == float(a[1]['theta']) and np.array_equal(a[0]['w'], a[1]['w'])     # replicated parameters
    b = _two_ranks(kind, mode, d, B, episodes, tmp_path, 'b')
    assert float(a[0]['theta']) == float(b[0]['theta']) and np.array_equal(a[0]['w'], b[0]['w'])     # run-to-run identical
    theta1, w1 = _run_class(kind, mode, d, B, episodes, host_seed=100, world=1)                       # rank 0's host seed
    assert float(a[0]['theta']) != (8.86349 if kind == 'ac' else 8.64)
    assert abs(float(a[0]['theta']) - theta1) <= 1e-12 * abs(theta1)
    assert np.max(np.abs(a[0]['w'] - w1)) <= 1e-12 * np.max(np.abs(w1))


# ---------------------------------------------------------------------------------------------------------------------
# Reward learning with two ranks (replicated reward network, ONE all-reduce of its flat gradient per update)
# ---------------------------------------------------------------------------------------------------------------------
def _irl_reward_learning(world_seed):
    """outerloop of AC_IRL (device D_samp, HIP training step) on this rank; returns (flat reward-net parameters, theta)."""
    sys.path.insert(0, ROOT)
    import random
    import torch
    from discrete_mean_field_game_amd.ac_irl import AC_IRL
    d = 15
    rs = np.random.RandomState(4)
    mat = rs.dirichlet(np.ones(d), size=6)
    demos = [[(rs.dirichlet(np.ones(d)), rs.dirichlet(np.ones(d), size=d)) for _ in range(15)] for _ in range(7)]
    np.random.seed(11); torch.manual_seed(11)               # identical initial weights on every rank
    ac = AC_IRL(d=d, pi0=mat, demonstrations=demos, batch=24, num_policies=2, seed=5, reg='dropout_l1l2', verbose=0)
    random.seed(world_seed)                                  # DIFFERENT host sampler states: _sync_host_sampler must align them
    ac.outerloop(num_iterations=2, num_gen_from_policy=3, max_reward_iterations=12, max_forward_episodes=2, final_training=False)
    return ac._trainer.flat.cpu().numpy().copy(), float(np.ravel(ac.theta)[0]), int(ac._trainer.step_count)


def _child_irl(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        flat, theta, steps = _irl_reward_learning(1000 + 31 * rank)
        np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), flat=flat, theta=theta, steps=steps)
    finally:
        dist.destroy_process_group()


def test_irl_reward_learning_two_ranks_stay_replicated(tmp_path):
    """Two ranks, different host `random` states: after the seed hand-shake of reward_iteration they draw the same batches,
    the gradient-only launch + all-reduce(mean) + mfg_reward_net_adam leaves bit-identical reward networks on both, and the
    forward solves that follow (sharded batch, replicated update) end at the same theta."""
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU')
    import torch.multiprocessing as mp
    out_dir = str(tmp_path / 'irl2')
    os.makedirs(out_dir)
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_child_irl, args=(r, 2, port, out_dir)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    for p in procs:
        if p.is_alive():
            p.terminate()
            pytest.fail('child rank did not finish')
        assert p.exitcode == 0
    a, b = [np.load(os.path.join(out_dir, 'rank%d.npz' % r)) for r in range(2)]
    assert int(a['steps']) == int(b['steps']) == 24
    assert np.array_equal(a['flat'], b['flat']) and float(a['theta']) == float(b['theta'])
    assert np.all(np.isfinite(a['flat']))


# ---------------------------------------------------------------------------------------------------------------------
# Assigning the critic weights while a multi-rank update is still pending
# ---------------------------------------------------------------------------------------------------------------------
def _run_w_assignment(world):
    """train() is cut short with the update of its last finished episode still PENDING (the deferred multi-rank update is
    applied by the NEXT rollout kernel; here that rollout never starts), then `ac.w = ...` replaces the weights, then
    training goes on.  The assignment must apply the pending increment to the OLD weights first (like the theta setter) --
    otherwise the next flush adds the stale increment onto the user's values.  Returns (theta, w) at the end."""
    sys.path.insert(0, ROOT)
    from discrete_mean_field_game_amd import ops
    from discrete_mean_field_game_amd.mfg_ac2 import actor_critic
    d, B = 21, 50
    np.random.seed(11)
    ac = actor_critic(d=d, pi0=_mat(d), batch=B, rng='philox', seed=5, update_every='rollout', precision='f64', verbose=0)
    new_w = np.random.RandomState(77).rand(ops.num_features(d))
    if world == 1:
        ac.train(num_episodes=2, gamma=0.9)
        assert ac._pending is None
    else:
        real, calls = ops.train_rollout_deferred, {'n': 0}

        def cut(*a, **k):
            calls['n'] += 1
            if calls['n'] == 3:
                raise RuntimeError('interrupted')            # episode 2 never starts; the update of episode 1 stays pending
            return real(*a, **k)
        ops.train_rollout_deferred = cut
        try:
            with pytest.raises(RuntimeError, match='interrupted'):
                ac.train(num_episodes=5, gamma=0.9)
        finally:
            ops.train_rollout_deferred = real
        assert ac._pending is not None
    ac.w = new_w
    assert ac._pending is None and np.array_equal(ac.w[:, 0], new_w)       # nothing left to be added onto the new weights
    ac.train(num_episodes=2, gamma=0.9, first_episode=2)
    return float(np.ravel(ac.theta)[0]), ac.w[:, 0].copy()


def _child_w(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        theta, w = _run_w_assignment(world)
        np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), theta=theta, w=w)
    finally:
        dist.destroy_process_group()


def test_assigning_w_with_a_pending_multi_rank_update(tmp_path):
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU')
    import torch.multiprocessing as mp
    out_dir = str(tmp_path / 'wset')
    os.makedirs(out_dir)
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_child_w, args=(r, 2, port, out_dir)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    for p in procs:
        if p.is_alive():
            p.terminate()
            pytest.fail('child rank did not finish')
        assert p.exitcode == 0
    a, b = [np.load(os.path.join(out_dir, 'rank%d.npz' % r)) for r in range(2)]
    assert float(a['theta']) == float(b['theta']) and np.array_equal(a['w'], b['w'])
    theta1, w1 = _run_w_assignment(1)
    assert abs(float(a['theta']) - theta1) <= 1e-12 * abs(theta1)
    assert np.max(np.abs(a['w'] - w1)) <= 1e-12 * np.max(np.abs(w1))
