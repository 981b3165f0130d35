"""Developer tool: time the wave-per-trajectory fused kernels (C3, C5 share; rollout with and without TD) on one GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from discrete_mean_field_game_amd import ops

dev = torch.device('cuda:0')
shapes = [(128, 40, 16384), (256, 40, 16384)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for d, T, B in shapes:
    th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(0)
    pi = torch.rand(B, d, device=dev, generator=g); pi = (pi / pi.sum(1, keepdim=True)).contiguous()
    F = ops.num_features(d)
    w = torch.rand(F, dtype=torch.float64, device=dev, generator=g)
    G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
    ws = ops.workspace(B * T, d, dev)
    for td in (True, False):
        out = ops.rollout(pi, T, th, 0.16, 12000.0, w=w if td else None, seed=1, td=td, G=G if td else None, ws=ws)
        def step():
            ops.rollout(pi, T, th, 0.16, 12000.0, w=w if td else None, seed=1, td=td, G=G if td else None, ws=ws, out=out)
        for _ in range(2): step()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 4
        for _ in range(n): step()
        e1.record(); torch.cuda.synchronize()
        print('d=%d T=%d B=%d %s  %.3f ms' % (d, T, B, 'training rollout (TD, sums)' if td else 'rollout only', e0.elapsed_time(e1) / n), flush=True)
    assert ops.status() == 0
