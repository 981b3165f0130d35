"""Developer probe (round 6), second form of hybrid_probe.py: ONE pair per measurement -- synchronise, launch the packed kernel over B1
trajectories on one stream and k_core_row3 over B2 on another, synchronise -- against each mapping alone over B1 + B2, timed the same
way (the launch + synchronise overhead is common to all three).  Median of many repetitions.  usage: hybrid_probe2.py [T B1 B2] ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops, _lib as L

dev = torch.device('cuda:0')
d = 21
lib = L.lib()
th = torch.tensor([8.86349], dtype=torch.float64, device=dev)


def setup(B, T):
    rs = np.random.RandomState(0)
    mat = torch.as_tensor(rs.dirichlet(np.ones(d), size=64).astype(np.float32), device=dev)
    pi = ops.gather_start(mat, torch.as_tensor(rs.randint(64, size=B).astype(np.int32), device=dev))
    w = torch.as_tensor(rs.rand(ops.num_features(d)), device=dev)
    out = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'reward': torch.empty(B, T, device=dev), 'pi_last': torch.empty(B, d, device=dev),
           'delta': torch.empty(B, T, dtype=torch.float64, device=dev), 'g': torch.empty(B, T, dtype=torch.float64, device=dev)}
    return pi, w, out, T


def launch(st, mode):
    pi, w, out, T = st
    lib.mfg_set_core_mapping(mode)
    ops.rollout(pi, T, th, 0.16, 12000.0, w=w, seed=1, td=True, reward_kind=2, out=out)


def med(fn, n=300):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6, ts[len(ts) // 10] * 1e6


a = [int(x) for x in sys.argv[1:]] or [15, 3072, 1024, 1, 3072, 1024]
for T, B1, B2 in zip(a[0::3], a[1::3], a[2::3]):
    s_all, s1, s2 = setup(B1 + B2, T), setup(B1, T), setup(B2, T)
    st1, st2 = torch.cuda.Stream(), torch.cuda.Stream()

    def pair():
        with torch.cuda.stream(st1):
            launch(s1, 1)
        with torch.cuda.stream(st2):
            launch(s2, 2)
    empty = med(lambda: None)
    p = med(lambda: launch(s_all, 1)); r = med(lambda: launch(s_all, 2)); h = med(pair)
    print('T = %2d, B = %d: median (10th percentile) us incl. launch + synchronise [empty: %.1f] -- packed alone %.1f (%.1f) | row3 alone %.1f (%.1f) | '
          'packed %d || row3 %d on two streams %.1f (%.1f)' % (T, B1 + B2, empty[0], p[0], p[1], r[0], r[1], B1, B2, h[0], h[1]), flush=True)
lib.mfg_set_core_mapping(0)
