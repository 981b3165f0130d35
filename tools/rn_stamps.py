"""Phase stamps of the matrix-core reward-network kernel (developer tool).
Build the instrumented library first:  bash tools/variant.sh rn_stamps mfg_reward_net.hip "-DMFG_RN_STAMPS"
then on the GPU box:  MFG_HIP_LIB=.../variants/librn_stamps.so python tools/rn_stamps.py [B]
Prints, for blocks 0 and 100 (first group), the shader-clock offsets of the phase boundaries of waves 0 .. 15."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd.ac_irl import AC_IRL
from discrete_mean_field_game_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rs = np.random.RandomState(0)
mat = rs.dirichlet(np.ones(21), size=64)
np.random.seed(5); torch.manual_seed(5)
ac = AC_IRL(theta=8.64, shift=0.0, alpha_scale=1e4, d=21, pi0=mat, demonstrations=[], batch=B, seed=3, update_every='step', verbose=0)
ac.train(max_episodes=3, stop_criteria=-1)
torch.cuda.synchronize()
lib = C.CDLL(os.environ['MFG_HIP_LIB'])
buf = (C.c_ulonglong * (2 * 16 * 16))()
assert lib.mfg_debug_rn_stamps(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(2, 16, 16).astype(np.int64)
names = ['entry', 'prologue issued', 'barrier 0 passed', 'setup done', 'conv1 done', 'conv2 + acts done', 'barrier a passed',
         'mfma + partials done', 'barrier b passed', 'reward stored', 'sums folded', 'kernel end', 'c1: tile written', 'c1: centre row done', 'c1: rows above done', '-']
for blk in range(2):
    t0 = t[blk, :, 0].min()
    print('block', (0, 100)[blk])
    for i, n in enumerate(names):
        print('  %-22s' % n, ' '.join('%6d' % (t[blk, w, i] - t0) for w in range(16)))
