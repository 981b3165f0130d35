"""-m gpu: the N > 1 legs of bench.py (`other_scaling`, `c5_strong`, `collective`) executed for real in fresh child processes:
two ranks share GPU 0 and talk over gloo (`--share-gpu --backend gloo`: the debug switches bench.py has for exactly this; RCCL
refuses two ranks on one GPU), launched like the driver launches the scaling bench (`python -m torch.distributed.run
--nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...`).  What is validated is the code path and the JSON contract of the
multi-GPU line, not a speed."""
import json
import os
import socket
import subprocess
import sys

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


def test_bench_two_ranks_prints_one_valid_line():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--share-gpu', '--backend', 'gloo', '--no-cpu-baseline', '--batch', '8192', '--c5-batch', '64']
    env = dict(os.environ, MFG_BENCH_HEAT='5', HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'rank 0 prints exactly ONE JSON line, got %d' % len(lines)
    z = json.loads(lines[0])
    assert z['n_gpus'] == 2 and z['steps'] == 2 and z['warmup'] == 1 and z['higher_is_better'] is True
    assert z['metric'].startswith('env-steps/sec') and z['unit'] == 'env-steps/s' and z['data'] == 'synthetic'
    assert z['scaling'] == 'strong' and z['config']['global_batch'] == 8192 and z['config']['batch_per_gpu'] == 4096
    assert z['value'] > 0 and abs(z['value'] - 8192 * 15 * 2 / (z['ms_per_step'] * 2e-3)) <= 1e-6 * z['value']
    assert z['cpu_baseline'] is None and z['vs_baseline'] is None
    o = z['other_scaling']
    assert o['scaling'] == 'weak' and o['batch_per_gpu'] == 8192 and o['value'] > 0
    c5 = z['c5_strong']
    assert c5['batch_per_gpu'] == 32 and c5['value'] > 0 and 'd=256 T=40' in c5['workload']
    col = z['collective']
    assert col['world'] == 2 and col['payload_bytes'] == (21 * 22 // 2 + 21 + 1 + 3) * 8 and col['all_reduce_us'] > 0
    assert z['roofline']['bound'] == 'hbm' and 0 < z['roofline']['frac'] < 1.2
    lp = z['loops']                                    # both episode loops timed in one run (gloo: the native RCCL loop cannot exist)
    assert lp['headline_loop'] == 'torch_dist' and lp['native_rccl_ms_per_update'] is None and lp['torch_dist_ms_per_update'] > 0
