"""Event-timed fused rollout kernels only (developer tool): TD rollout (+gradient kernel), env-only rollout.
usage: core_probe.py [f64] [d,B,T ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops

dev = torch.device('cuda:0')

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

def probe(d, B, T, prec='mixed'):
    th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
    rs = np.random.RandomState(0)
    mat = rs.dirichlet(np.ones(d), size=64)
    pi = torch.as_tensor(mat[rs.randint(64, size=B)].astype(np.float32), device=dev)
    w = torch.as_tensor(rs.rand(ops.num_features(d)), device=dev)
    ws = ops.workspace(B * T, d, dev)
    out = ops.rollout(pi, T, th, 0.16, 12000.0, w=w, seed=1, td=True, ws=ws, precision=prec)
    n = 10 if d <= 64 else 3
    t = timeit(lambda: ops.rollout(pi, T, th, 0.16, 12000.0, w=w, seed=1, td=True, ws=ws, out=out, G=out['G'], precision=prec), n=n, warm=2)
    print('d=%d B=%d T=%d %s: TD rollout %.3f ms  %.3e env-steps/s' % (d, B, T, prec, t * 1e3, B * T / t), flush=True)
    out2 = ops.rollout(pi, T, th, 0.16, 12000.0, seed=1, td=False, precision=prec)
    t = timeit(lambda: ops.rollout(pi, T, th, 0.16, 12000.0, seed=1, td=False, out=out2, precision=prec), n=n, warm=2)
    print('      env-only rollout %.3f ms  %.3e env-steps/s' % (t * 1e3, B * T / t), flush=True)

if __name__ == '__main__':
    cfgs = [(21, 65536, 15), (21, 4096, 15), (128, 16384, 40), (256, 16384, 40)]
    prec = 'mixed'
    args = [a for a in sys.argv[1:] if a not in ('f64', 'mixed')]
    if 'f64' in sys.argv[1:]: prec = 'f64'
    if args:
        cfgs = [tuple(int(x) for x in a.split(',')) for a in args]
    for c in cfgs:
        probe(*c, prec=prec)
