"""The C ABI from a plain-C host program (examples/abi_consumer.c): compiles and links with gcc -std=c99 on CPU;
on the GPU its two training updates reproduce the Python binding's numbers bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_c_consumer_compiles_as_c99():
    import __graft_entry__ as ge
    exe = ge.build_c_consumer()
    assert os.access(exe, os.X_OK)
    # the header itself is C: no C++ tokens leak through the include
    src = open(os.path.join(ROOT, 'examples', 'abi_consumer.c')).read()
    assert '#include "mfg_hip.h"' in src and '#include <torch' not in src and 'extern "C"' not in src


@pytest.mark.gpu
def test_c_consumer_matches_python_binding():
    torch = pytest.importorskip('torch')
    if not torch.cuda.is_available():
        pytest.fail('-m gpu tests need a GPU: the HIP path has no CPU fallback')
    import __graft_entry__ as ge
    from discrete_mean_field_game_amd import ops
    exe = ge.build_c_consumer()
    B, d, T, num_start = 1000, 21, 15, 8
    res = subprocess.run([exe, str(B)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    got = json.loads(res.stdout.strip().splitlines()[-1])
    assert got['abi'] == ops.L.lib().mfg_abi_version() and got['B'] == B and got['count'] == B * T
    assert got['arch'].startswith('gfx950')
    # the same two updates through the ctypes binding
    dev = torch.device('cuda:0')
    mat = np.array([[(s * 31 + j * 17) % 97 + 1 for j in range(d)] for s in range(num_start)], dtype=np.float64)
    mat = (mat / mat.sum(1, keepdims=True)).astype(np.float32)
    idx = ((np.arange(B) * 7 + 3) % num_start).astype(np.int32)
    F = ops.num_features(d)
    w = torch.as_tensor(((np.arange(F) * 13) % 101) / 101.0, device=dev)
    theta = torch.tensor([8.86349], dtype=torch.float64, device=dev)
    racc = torch.zeros(1, dtype=torch.float64, device=dev)
    G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
    ws = ops.workspace(B * T, d, dev)
    for ep in range(2):
        pi0 = ops.gather_start(torch.as_tensor(mat, device=dev), torch.as_tensor(idx, device=dev))
        ops.rollout(pi0, T, theta, 0.16, 12000.0, w=w, gamma=1.0, seed=42, first_step=ep * T, td=True, G=G, ws=ws)
        ops.apply_update(G, d, 0.1 / (ep + 1), 0.001 / (ep + 1), w, theta, racc)
    assert got['theta'] == float(theta[0])
    assert got['mean_reward_acc'] == float(racc[0])
    assert abs(got['w_sum'] - float(w.sum())) <= 1e-12 * abs(float(w.sum()))
    # ... and the three further episodes the C program ran through the native episode loop (mfg_train_rollouts: start
    # states drawn on the device, learning-rate schedule evaluated in C) against the per-episode Python sequence
    from discrete_mean_field_game_amd.parallel import lr_scales
    matd = torch.as_tensor(mat, device=dev)
    bufs = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'pi_last': torch.empty(B, d, device=dev),
            'reward': torch.empty(B, T, device=dev), 'delta': torch.empty(B, T, dtype=torch.float64, device=dev),
            'g': torch.empty(B, T, dtype=torch.float64, device=dev)}
    racc3 = torch.zeros(3, dtype=torch.float64, device=dev)
    for k in range(3):
        sc, sa = lr_scales(2 + k, False)
        ops.train_rollout(matd, None, T, theta, 0.16, 12000.0, w, 1.0, G, ws, bufs, 0.1 * sc, 0.001 * sa, apply=True, seed=42,
                          first_step=(2 + k) * T, reward_acc=racc3[k:k + 1])
    assert got['theta_after_native_loop'] == float(theta[0])
    assert got['native_loop_rewards'] == [float(v) for v in racc3.cpu()]
