"""Measure every BASELINE.json config shape on one GPU (per-GPU share for C5) and print a markdown table.
Legs: fused training rollout (sample+transition+reward+value+TD+score+batch sums+update) and the HBM-bound
given-P transition+reward kernel on a slab of ~1.6 GB (or the config's own B*T transitions if smaller)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops

dev = torch.device('cuda:0')

def timeit(fn, n, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

rows = []
for name, d, T, B in (('C2', 21, 15, 4096), ('target', 21, 15, 65536), ('C3', 128, 40, 16384),
                      ('C5 (1/8 share)', 256, 40, 16384)):
    th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(0)
    pi = torch.rand(B, d, device=dev, generator=g); pi = (pi / pi.sum(1, keepdim=True)).contiguous()
    F = ops.num_features(d)
    w = torch.rand(F, dtype=torch.float64, device=dev, generator=g)
    G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
    ws = ops.workspace(B * T, d, dev)
    out = ops.rollout(pi, T, th, 0.16, 12000.0, w=w, seed=1, td=True, G=G, ws=ws)
    def step():
        ops.rollout(pi, T, th, 0.16, 12000.0, w=w, seed=1, td=True, G=G, ws=ws, out=out)
        ops.apply_update(G, d, 1e-9, 1e-9, w, th)
    t = timeit(step, 3 if d > 64 else 10)
    bytes_step = 4 * (d * d + 2 * d + 1)
    N = max(B, int(1.6e9 // (4 * d * d)))
    piB = torch.rand(N, d, device=dev, generator=g)
    PB = torch.rand(N, d, d, device=dev, generator=g)
    tg = timeit(lambda: ops.step_given_P(piB, PB), 10, 2)
    rows.append('| %s | %d | %d | %d | %.2f ms | %.3g | %.3g | %.2f TB/s (%.0f %%) |' % (
        name, d, T, B, t * 1e3, B * T / t, N / tg, N * bytes_step / tg / 1e12, 100 * N * bytes_step / tg / 8e12))
    del piB, PB, out
print('| config | d | T | B | fused train rollout | env-steps/s (fused) | steps/s (given-P) | given-P HBM |')
print('|---|---|---|---|---|---|---|---|')
print('\n'.join(rows))
