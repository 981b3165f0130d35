"""s_memtime stamps of one wave of the small-d fused kernel (needs the -DMFG_TIMING variant library; developer tool).
usage: MFG_HIP_LIB=.../libtiming.so python tools/phase_timing.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
dev = torch.device('cuda:0')
buf = torch.zeros(64, dtype=torch.int64, device=dev)
os.environ['MFG_TIMING_BUF'] = '%x' % buf.data_ptr()
from discrete_mean_field_game_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d, T = 21, 15
th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
rs = np.random.RandomState(0)
pi0 = torch.as_tensor(rs.dirichlet(np.ones(d), size=B).astype(np.float32), device=dev)
w = torch.as_tensor(rs.rand(ops.num_features(d)), device=dev)
for td in (True, False):
    for _ in range(3):
        ops.rollout(pi0, T, th, 0.16, 12000.0, w=w if td else None, seed=7, td=td)
    torch.cuda.synchronize()
    s = buf.cpu().numpy().reshape(4, 16)
    names = ['stage state/E/F', 'quad loop (sampling)', 'normalise + row epilogue', 'column pass', 'sums + value + outputs']
    print('B=%d td=%s   (s_memtime ticks: 100 MHz constant clock -> x21 for shader cycles at 2.1 GHz)' % (B, td))
    for step in range(1, 4):
        row = s[step]
        print('  step %d: ' % step + ', '.join('%s %d' % (names[k], row[k + 1] - row[k]) for k in range(5)) + ', total %d' % (row[5] - row[0]))
