"""-m gpu: the RCCL leg of the multi-GPU path, executed for real on the one GPU a test box has.

`north_star` / SURVEY.md 8e: one process per GPU, ONE RCCL all-reduce of the fused gradient buffer per update.  The
world-size-2 tests (tests/test_parallel_gloo.py on the CPU, tests/test_gpu_multiproc.py on the GPU) run over gloo, because
two ranks cannot share one device under RCCL.  What they leave unexecuted is the "nccl" branch itself:
`init_process_group('nccl', device_id=...)`, a device-tensor `all_reduce` of G, `mfg_apply_update` after it, the barrier +
max-over-ranks timing of bench.py.  Here a FRESH child process (started before it touches the GPU) runs

    bench.py --gpus 1 --force-dist ...     -> the N > 1 code path of bench.py with a 1-rank RCCL communicator
    actor_critic / AC_IRL .train(group=WORLD) with backend nccl

and the parent checks the exit code, the JSON line and the trained parameters against the single-process run.
"""
import json
import os
import socket
import subprocess
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


def _env():
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', LOCAL_RANK='0', WORLD_SIZE='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    env['PYTHONPATH'] = ROOT + os.pathsep + env.get('PYTHONPATH', '')
    return env


def test_bench_force_dist_runs_the_rccl_all_reduce():
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-dist', '--steps', '2', '--warmup', '1',
           '--no-configs', '--no-cpu-baseline', '--no-roofline', '--batch', '8192']
    p = subprocess.run(cmd, cwd=ROOT, env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1]
    z = json.loads(line)
    assert z['n_gpus'] == 1 and np.isfinite(z['value']) and z['value'] > 0 and np.isfinite(z['theta_end'])
    c = z['collective']
    assert c['backend'].startswith('rccl') and c['payload_bytes'] == (253 + 3) * 8
    assert 0 < c['all_reduce_us'] < 5e3 and 0 < c['all_reduce_plus_apply_update_us'] < 5e3
    lp = z['loops']                                    # the native loop enabled itself behind its canary; both loops are timed
    assert lp['headline_loop'] == 'native_rccl' and lp['canary']['native'] and lp['canary']['stage'] == 'ok'
    assert lp['native_rccl_ms_per_update'] > 0 and lp['torch_dist_ms_per_update'] > 0 and c['native_all_reduce_us'] > 0
    print('RCCL 1-rank all-reduce of G: %.1f us; + mfg_apply_update: %.1f us' % (c['all_reduce_us'],
                                                                                 c['all_reduce_plus_apply_update_us']))


_CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from discrete_mean_field_game_amd import parallel
from discrete_mean_field_game_amd.mfg_ac2 import actor_critic
# (1) the collective on a device tensor, through the product's own helper -- a 1-rank communicator is still an RCCL call
calls = {'n': 0}
real = dist.all_reduce
def counted(t, *a, **k):
    calls['n'] += 1
    assert t.is_cuda
    return real(t, *a, **k)
dist.all_reduce = counted
parallel.dist.get_world_size = lambda group=None: 2          # make the helpers take their multi-rank branches ...
parallel.current_shard = lambda B, group=None: parallel.shard_batch(B, 0, 1)   # ... while this rank owns the whole batch
import discrete_mean_field_game_amd.mfg_ac2 as M
M.current_shard = parallel.current_shard
G = torch.arange(8, dtype=torch.float64, device='cuda')
parallel.all_reduce_gradients_(G)
assert calls['n'] == 1 and torch.equal(G.cpu(), torch.arange(8, dtype=torch.float64))
idx = parallel.broadcast_start_indices(np.arange(5), None, torch.device('cuda', 0))
assert list(idx) == [0, 1, 2, 3, 4]
dist.all_reduce = real
torch.cuda.synchronize()
print(json.dumps({'ok': True, 'all_reduce_calls': calls['n'], 'backend': dist.get_backend()}))
dist.destroy_process_group()
'''


def test_product_helpers_over_an_rccl_communicator():
    """parallel.all_reduce_gradients_ / broadcast_start_indices on DEVICE tensors over backend 'nccl' (their production
    branch), in a fresh child."""
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU')
    p = subprocess.run([sys.executable, '-c', _CHILD % {'root': ROOT}], cwd=ROOT, env=_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    z = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert z['ok'] and z['backend'] == 'nccl' and z['all_reduce_calls'] == 1


_CHILD_NATIVE = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from discrete_mean_field_game_amd import parallel, ops, _lib as L
from discrete_mean_field_game_amd.mfg_ac2 import actor_critic
import ctypes as C
mat = np.random.RandomState(3).dirichlet(np.ones(21), size=7)
res = {}
# the library's own communicator (a job with several ranks makes it in the class constructor; with ONE rank it only exists
# for this test): bootstrapped through the process group, cached per group
comm = parallel.native_comm(None, torch.device('cuda', 0), allow_single=True)
calls = {'train_rollouts_dist': 0, 'all_reduce': 0}
real_d, real_ar = ops.train_rollouts_dist, dist.all_reduce
ops.train_rollouts_dist = lambda *a, **k: (calls.__setitem__('train_rollouts_dist', calls['train_rollouts_dist'] + 1), real_d(*a, **k))[1]
for mode in ('native_dist', 'single'):
    np.random.seed(11)
    ac = actor_critic(d=21, pi0=mat, batch=500, rng='philox', seed=5, update_every='rollout', verbose=0)
    ac._force_collective = mode == 'native_dist'
    logs = []
    ac.train_log = lambda v, f, fmt, logs=logs: logs.append(np.array(v, dtype=np.float64).tolist())
    if mode == 'native_dist':
        dist.all_reduce = lambda *a, **k: (_ for _ in ()).throw(AssertionError('train() must not go through torch.distributed per episode'))
    ac.train(num_episodes=7, gamma=0.9, constant=0, consecutive=3, write_file=1, first_episode=1)
    dist.all_reduce = real_ar
    res[mode] = (float(np.ravel(ac.theta)[0]), ac.w[:, 0].tolist(), logs, ac._rng_step)
# the exchange itself through the library's communicator: a 1-rank SUM is the identity
assert parallel.native_comm(None, torch.device('cuda', 0), allow_single=True) == comm      # cached
G = torch.arange(8, dtype=torch.float64, device='cuda')
L.check(L.lib().mfg_dist_all_reduce(comm, G.data_ptr(), 8, torch.cuda.current_stream().cuda_stream), 'mfg_dist_all_reduce')
torch.cuda.synchronize()
print(json.dumps({'same_theta': res['native_dist'][0] == res['single'][0], 'same_w': res['native_dist'][1] == res['single'][1],
                  'same_logs': res['native_dist'][2] == res['single'][2], 'steps': [res['native_dist'][3], res['single'][3]],
                  'dist_calls': calls['train_rollouts_dist'], 'comm': bool(comm), 'G': G.cpu().tolist(),
                  'theta': res['native_dist'][0]}))
dist.destroy_process_group()
'''


def test_native_rccl_episode_loop_equals_the_single_gpu_loop():
    """mfg_train_rollouts_dist: the multi-GPU episode loop with the all-reduce issued by the library itself (its own RCCL
    communicator, bootstrapped through the process group), on a 1-rank communicator: the class takes it (no per-episode
    torch.distributed call), chunks it at the reports like the single-GPU native loop, and -- a 1-rank sum being the
    identity and the deferred update being bit-equal to an update launch -- ends with the same parameters, reports and
    Philox counter as the single-GPU run."""
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU')
    # (round 6: the native loop enables itself behind a canary, parallel.native_comm -- no environment switch)
    p = subprocess.run([sys.executable, '-c', _CHILD_NATIVE % {'root': ROOT}], cwd=ROOT, env=_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    z = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert z['comm'] and z['dist_calls'] == 3                      # chunks: episode 0 | 1..3 | 4..6 (reports at 0, 3, 6)
    assert z['same_theta'] and z['same_w'] and z['same_logs'] and z['steps'] == [105, 105]
    assert z['G'] == list(range(8)) and z['theta'] != 8.86349


_CHILD_CANARY = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from discrete_mean_field_game_amd import parallel, ops, _lib as L
from discrete_mean_field_game_amd.mfg_ac2 import actor_critic
mat = np.random.RandomState(3).dirichlet(np.ones(21), size=7)
comm = parallel.native_comm(None, torch.device('cuda', 0), allow_single=True)
log = dict(parallel.CANARY_LOG[-1])
calls = {'dist': 0, 'all_reduce': 0}
real_d, real_ar = ops.train_rollouts_dist, dist.all_reduce
ops.train_rollouts_dist = lambda *a, **k: (calls.__setitem__('dist', calls['dist'] + 1), real_d(*a, **k))[1]
dist.all_reduce = lambda *a, **k: (calls.__setitem__('all_reduce', calls['all_reduce'] + 1), real_ar(*a, **k))[1]
res = {}
for mode in ('forced', 'single'):
    np.random.seed(11)
    ac = actor_critic(d=21, pi0=mat, batch=500, rng='philox', seed=5, update_every='rollout', verbose=0)
    ac._force_collective = mode == 'forced'
    ac.train(num_episodes=5, gamma=0.9)
    res[mode] = (float(np.ravel(ac.theta)[0]), ac.w[:, 0].tolist())
torch.cuda.synchronize()
print(json.dumps({'comm': bool(comm), 'log': log, 'calls': calls, 'cached_none': parallel.native_comm(None, torch.device('cuda', 0), allow_single=True) is None,
                  'same': res['forced'] == res['single'], 'theta': res['forced'][0]}))
dist.destroy_process_group()
'''


@pytest.mark.parametrize('inject', ['', 'all', 'off'])
def test_native_rccl_canary_enables_the_loop_or_falls_back_symmetrically(inject):
    """the native RCCL loop is self-enabling behind a canary (parallel.native_comm: agreed ncclCommInitRank,
    then one patterned all-reduce + a tiny mfg_train_rollouts_dist in a helper thread, then a final agreement).  1-rank `nccl`
    communicator, fresh child: a healthy canary switches the class to the native loop (no torch.distributed call per episode);
    an injected canary failure (MFG_NATIVE_RCCL_CANARY_FAIL) makes the rank abort its -- healthy -- communicator and stay on
    torch.distributed, with the same trained parameters; MFG_NATIVE_RCCL=0 switches the attempt off."""
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU')
    env = _env()
    if inject == 'all':
        env['MFG_NATIVE_RCCL_CANARY_FAIL'] = 'all'
    if inject == 'off':
        env['MFG_NATIVE_RCCL'] = '0'
    p = subprocess.run([sys.executable, '-c', _CHILD_CANARY % {'root': ROOT}], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    z = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert z['same'] and z['theta'] != 8.86349
    if inject == '':
        assert z['comm'] and z['log']['native'] and z['log']['stage'] == 'ok' and not z['cached_none']
        assert z['calls']['dist'] >= 1                      # the class took the native loop ...
    else:
        assert not z['comm'] and not z['log']['native'] and z['cached_none']
        assert z['log']['stage'] == ('canary' if inject == 'all' else 'resolve')
        if inject == 'all':
            assert 'injected' in z['log']['detail']
        assert z['calls']['dist'] == 0 and z['calls']['all_reduce'] >= 5     # ... or one torch all-reduce per update


_CHILD_NATIVE_LARGE = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from discrete_mean_field_game_amd import parallel, ops
from discrete_mean_field_game_amd.mfg_ac2 import actor_critic
res = {}
calls = {'n': 0}
real_d = ops.train_rollouts_dist
ops_patch = lambda *a, **k: (calls.__setitem__('n', calls['n'] + 1), real_d(*a, **k))[1]
for d, B, T in ((128, 24, 3), (256, 10, 2), (100, 9, 2)):
    mat = np.random.RandomState(3).dirichlet(np.ones(d), size=5)
    out = []
    for mode in ('native_dist', 'torch_dist', 'single'):
        np.random.seed(11)
        ac = actor_critic(d=d, pi0=mat, batch=B, rng='philox', seed=5, update_every='rollout', verbose=0, episode_steps=T)
        ac._force_collective = mode != 'single'
        ac._use_native_rccl = mode == 'native_dist'
        ops.train_rollouts_dist = ops_patch
        ac.train(num_episodes=4, gamma=0.9)
        ops.train_rollouts_dist = real_d
        out.append((float(np.ravel(ac.theta)[0]), ac.w[:, 0].tolist()))
    res[d] = {'native_eq_single': out[0] == out[2], 'torch_eq_single': out[1] == out[2], 'theta': out[0][0]}
torch.cuda.synchronize()
print(json.dumps({'res': res, 'native_calls': calls['n']}))
dist.destroy_process_group()
'''


def test_native_rccl_loop_at_large_d_equals_the_single_gpu_run():
    """The c5_strong leg of bench.py (d = 256) takes the native RCCL loop on a multi-GPU node: mfg_train_rollouts_dist with the
    wave-per-trajectory kernels (the deferred update is an out-of-place launch in front of the rollout there).
    1-rank `nccl` communicator, fresh child: d = 128 / 256 / 100 through the class, native loop and torch.distributed loop both
    bit-equal to the single-GPU run."""
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU')
    p = subprocess.run([sys.executable, '-c', _CHILD_NATIVE_LARGE % {'root': ROOT}], cwd=ROOT, env=_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    z = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith('{')][-1])
    assert z['native_calls'] in (3, 4)       # one native call per (silent) train() of the three shapes (+ the canary's own, made on first use)
    for d, r in z['res'].items():
        assert r['native_eq_single'] and r['torch_eq_single'] and r['theta'] != 8.86349, (d, r)
