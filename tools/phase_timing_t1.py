"""s_memtime stamps of one wave of a T = 1 launch of the small-d fused kernel (timing variant library): prologue (weight
staging, V of the start state), the step phases, and the tail, with and without the P copy-out.
usage: MFG_HIP_LIB=.../libtiming.so python tools/phase_timing_t1.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
dev = torch.device('cuda:0')
buf = torch.zeros(64 + 2 * 4096, dtype=torch.int64, device=dev)
os.environ['MFG_TIMING_BUF'] = '%x' % buf.data_ptr()
from discrete_mean_field_game_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = 21
th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
rs = np.random.RandomState(0)
pi0 = torch.as_tensor(rs.dirichlet(np.ones(d), size=B).astype(np.float32), device=dev)
w = torch.as_tensor(rs.rand(ops.num_features(d)), device=dev)
for wp in (False, True):
    for _ in range(3):
        ops.rollout(pi0, 1, th, 0.16, 12000.0, w=w, seed=7, td=True, write_P=wp, reward_kind=2 if wp else 0)
    torch.cuda.synchronize()
    allb = buf.cpu().numpy()
    s = allb[:64].reshape(4, 16)
    nb = (B + 11) // 12
    be = allb[64:64 + 2 * nb].reshape(nb, 2)
    t0 = be[:, 0].min()
    print('  blocks: entry %d .. %d, exit %d .. %d ticks after the first entry; block lifetime %d .. %d (median %d)'
          % (0, be[:, 0].max() - t0, be[:, 1].min() - t0, be[:, 1].max() - t0, (be[:, 1] - be[:, 0]).min(), (be[:, 1] - be[:, 0]).max(), int(np.median(be[:, 1] - be[:, 0]))))
    r = s[0]
    print('B=%d T=1 write_P=%s: entry->weights staged %d, ->V(start) done %d, step: stage %d quad %d epilogue %d column %d sums+value+out %d, tail %d; total %d ticks'
          % (B, wp, r[9] - r[8], r[10] - r[9], r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[11] - r[5], r[11] - r[8]))
