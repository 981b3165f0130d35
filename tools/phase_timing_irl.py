"""s_memtime stamps (10 ns ticks) of block 0 / wave 0 of the IRL step kernel with the reward network inside (k_core_small<..., RN>,
T = 1 launches of mfg_train_episode_irl): prologue, step phases, the reward-network phase, tail; plus every block's lifetime.
usage: MFG_HIP_LIB=.../libtiming.so python tools/phase_timing_irl.py [B]   (timing variant: tools/build_timing.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
dev = torch.device('cuda:0')
buf = torch.zeros(64 + 2 * 8192, dtype=torch.int64, device=dev)
os.environ['MFG_TIMING_BUF'] = '%x' % buf.data_ptr()
from discrete_mean_field_game_amd.ac_irl import AC_IRL
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = 21
rs = np.random.RandomState(0)
mat = rs.dirichlet(np.ones(d), size=64)
ac = AC_IRL(theta=8.64, shift=0.0, alpha_scale=1e4, d=d, pi0=mat, demonstrations=[], batch=B, seed=3, update_every='step', verbose=0)
ac.train(max_episodes=3, stop_criteria=-1)
torch.cuda.synchronize()
allb = buf.cpu().numpy()
s = allb[:64].reshape(4, 16)
nb = (B + 11) // 12
be = allb[64:64 + 2 * nb].reshape(nb, 2)
t0 = be[:, 0].min()
print('blocks: entry 0 .. %d, exit %d .. %d ticks after the first entry; lifetime %d .. %d (median %d)'
      % (be[:, 0].max() - t0, be[:, 1].min() - t0, be[:, 1].max() - t0, (be[:, 1] - be[:, 0]).min(), (be[:, 1] - be[:, 0]).max(),
         int(np.median(be[:, 1] - be[:, 0]))))
r = s[0]
print('B=%d last launch, block 0 wave 0 (10 ns ticks): entry->weights staged %d, ->tile loop %d, step: stage %d quad %d epilogue %d column %d '
      'reward net %d sums+value+out %d, tail %d; total %d' % (B, r[9] - r[8], r[10] - r[9], r[1] - r[0], r[2] - r[1], r[3] - r[2], r[6] - r[3],
                                                           r[4] - r[6], r[5] - r[4], r[11] - r[5], r[11] - r[8]))
