"""Phase stamps of the row-mapped reward-network kernel (developer tool): second pass of blocks 0 and 100.
Build the instrumented library first:  MFG_VARIANT_DIR=discrete_mean_field_game_amd/csrc/ab bash tools/variant.sh rn_stamps mfg_reward_net.hip "-DMFG_RN_STAMPS"
then on the GPU box:  MFG_HIP_LIB=.../ab/librn_stamps.so python tools/rr_stamps.py [B] [reg]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops
from discrete_mean_field_game_amd.networks import RewardNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
reg = sys.argv[2] if len(sys.argv) > 2 else 'dropout_l1l2'
dev = torch.device('cuda:0')
torch.manual_seed(0)
d = 21
net = RewardNet(d=d, reg=reg).to(dev)
rs = np.random.RandomState(0)
s = torch.as_tensor(rs.dirichlet(np.ones(d), size=B).astype(np.float32), device=dev)
a = torch.as_tensor(rs.dirichlet(np.ones(d), size=(B, d)).astype(np.float32), device=dev)
for _ in range(3): ops.reward_net_forward(net, s, a, seed=1)
torch.cuda.synchronize()
lib = C.CDLL(os.environ['MFG_HIP_LIB'])
buf = (C.c_ulonglong * (2 * 16 * 16))()
assert lib.mfg_debug_rn_stamps(buf) == 0
t = np.array(buf, dtype=np.uint64).reshape(2, 16, 16).astype(np.int64)
names = ['pass top', 'staged data landed', 'rows read, next stage issued', 'conv1 done', 'conv2 + acts written', 'barrier A passed',
         'mfma + partials', 'barrier B passed', 'tail done']
for blk in range(2):
    t0 = t[blk, :4, 0].min()
    print('block', (0, 100)[blk])
    for i, n in enumerate(names):
        print('  %-30s' % n, ' '.join('%6d' % (t[blk, w, i] - t0) for w in range(4)))
