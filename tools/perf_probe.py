"""Quick per-kernel timing probe (developer tool, not part of the bench contract)."""
import sys, os, time
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

def probe(d, B, T):
    th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(0)
    pi = torch.rand(B, d, device=dev, generator=g); pi = (pi / pi.sum(1, keepdim=True)).contiguous()
    F = ops.num_features(d)
    w = torch.rand(F, dtype=torch.float64, device=dev, generator=g)
    P = ops.sample_dirichlet(pi, th, 0.16, 12000.0, seed=1)
    bytes_step = 4 * (d * d + 2 * d + 1)
    t = timeit(lambda: ops.step_given_P(pi, P))
    print('d=%d B=%d step_given_P: %.1f us  %.3e steps/s  %.2f TB/s (%.1f%% of 8TB/s)' % (d, B, t*1e6, B/t, B*bytes_step/t/1e12, 100*B*bytes_step/t/8e12))
    # HBM leg: a slab larger than the 256 MiB L3
    Nbig = max(B, int(1.6e9 // (4 * d * d)))
    piB = torch.rand(Nbig, d, device=dev, generator=g); piB = (piB / piB.sum(1, keepdim=True)).contiguous()
    PB = torch.rand(Nbig, d, d, device=dev, generator=g)
    t = timeit(lambda: ops.step_given_P(piB, PB))
    print('   step_given_P on %.2f GB slab: %.1f us  %.3e steps/s  %.2f TB/s (%.1f%% of 8TB/s)' % (Nbig*d*d*4/1e9, t*1e6, Nbig/t, Nbig*bytes_step/t/1e12, 100*Nbig*bytes_step/t/8e12))
    del piB, PB
    t = timeit(lambda: ops.sample_dirichlet(pi, th, 0.16, 12000.0, seed=1, out=P))
    print('   sample_dirichlet: %.1f us  %.3e steps/s' % (t*1e6, B/t))
    pn, r = ops.step_given_P(pi, P)
    ws = ops.workspace(B * T, d, dev)
    t = timeit(lambda: ops.td_pg_accumulate(pi, pn, P, r, w, th, 0.16, 1.0, ws=ws))
    print('   td_pg_accumulate: %.1f us  %.3e steps/s' % (t*1e6, B/t))
    out = ops.rollout(pi, T, th, 0.16, 12000.0, w=w, seed=1, td=True, ws=ws)
    t = timeit(lambda: ops.rollout(pi, T, th, 0.16, 12000.0, w=w, seed=1, td=True, ws=ws, out=out, G=out['G']), n=5, warm=1)
    print('   rollout TD T=%d: %.2f ms  %.3e env-steps/s' % (T, t*1e3, B*T/t))
    out2 = ops.rollout(pi, T, th, 0.16, 12000.0, seed=1, td=False)
    t = timeit(lambda: ops.rollout(pi, T, th, 0.16, 12000.0, seed=1, td=False, out=out2), n=5, warm=1)
    print('   rollout env-only T=%d: %.2f ms  %.3e env-steps/s' % (T, t*1e3, B*T/t))

if __name__ == '__main__':
    cfgs = [(21, 65536, 15), (21, 4096, 15), (128, 16384, 4), (256, 4096, 2)]
    if len(sys.argv) > 1:
        cfgs = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
    for c in cfgs:
        probe(*c)
