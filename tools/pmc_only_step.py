"""rocprofv3 --pmc driver: only the given-P kernel (and its inputs) at a chosen shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops
d, N = 21, 983040
if len(sys.argv) > 1:
    d, N = (int(x) for x in sys.argv[1].split(','))
dev = torch.device('cuda:0')
g = torch.Generator(device=dev); g.manual_seed(0)
pi = torch.rand(N, d, device=dev, generator=g); pi = (pi / pi.sum(1, keepdim=True)).contiguous()
P = torch.rand(N, d, d, device=dev, generator=g); P = (P / P.sum(2, keepdim=True)).contiguous()
for _ in range(3):
    ops.step_given_P(pi, P)
torch.cuda.synchronize()
