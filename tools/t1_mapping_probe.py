"""Developer probe (round 6): ONE env step per launch (the shape of the per-step-update paths, C2 / C4 step mode) in both lane
mappings of the d = 21 sampling kernels -- plain mfg_rollout(T = 1) with TD outputs, with and without the actions written out.
(The SUMS / STEP variants of the per-step paths keep the packed mapping; this probe is the evidence for that choice.)
usage: t1_mapping_probe.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops, _lib as L

dev = torch.device('cuda:0')
lib = L.lib()
d = 21
th = torch.tensor([8.86349], dtype=torch.float64, device=dev)


def timeit(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for B in ([int(x) for x in sys.argv[1:]] or [4096, 2048, 1024]):
    rs = np.random.RandomState(0)
    pi = torch.as_tensor(rs.dirichlet(np.ones(d), size=B).astype(np.float32), device=dev)
    w = torch.as_tensor(rs.rand(ops.num_features(d)), device=dev)
    row = 'B = %5d, T = 1:' % B
    for wp in (False, True):
        out = ops.rollout(pi, 1, th, 0.16, 12000.0, w=w, seed=1, td=True, reward_kind=2, write_P=wp)
        for mode, name in ((1, 'packed'), (2, 'row3')):
            lib.mfg_set_core_mapping(mode)
            t = timeit(lambda: ops.rollout(pi, 1, th, 0.16, 12000.0, w=w, seed=1, td=True, reward_kind=2, write_P=wp, out=out))
            row += '  %s%s %5.2f us' % (name, ' +P' if wp else '', t)
    lib.mfg_set_core_mapping(0)
    print(row, flush=True)
