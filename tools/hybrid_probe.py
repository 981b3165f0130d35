"""Developer probe (round 6): would a HYBRID launch -- packed waves for most of a d = 21 batch, one-trajectory-per-wave waves for
the rest, side by side on every SIMD -- beat either mapping alone?  Emulated with two streams: the packed kernel over B1
trajectories on one, the row3 kernel over B2 on the other, paired per iteration with events (the cross-stream waits cost a few us:
an upper bound of what one hybrid launch would take).  usage: hybrid_probe.py [B1 B2] ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops, _lib as L

dev = torch.device('cuda:0')
d, T = 21, 15
lib = L.lib()


def setup(B):
    rs = np.random.RandomState(0)
    mat = torch.as_tensor(rs.dirichlet(np.ones(d), size=64).astype(np.float32), device=dev)
    pi = ops.gather_start(mat, torch.as_tensor(rs.randint(64, size=B).astype(np.int32), device=dev))
    w = torch.as_tensor(rs.rand(ops.num_features(d)), device=dev)
    out = {'pi_traj': torch.empty(B, T + 1, d, device=dev), 'reward': torch.empty(B, T, device=dev), 'pi_last': torch.empty(B, d, device=dev),
           'delta': torch.empty(B, T, dtype=torch.float64, device=dev), 'g': torch.empty(B, T, dtype=torch.float64, device=dev)}
    return pi, w, out


th = torch.tensor([8.86349], dtype=torch.float64, device=dev)


def launch(state, mode):
    pi, w, out = state
    lib.mfg_set_core_mapping(mode)
    ops.rollout(pi, T, th, 0.16, 12000.0, w=w, seed=1, td=True, reward_kind=2, out=out)


def alone(B, mode, n=40):
    st = setup(B)
    for _ in range(5):
        launch(st, mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        launch(st, mode)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def pair(B1, B2, n=40):
    a, b = setup(B1), setup(B2)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def it():
        ev = torch.cuda.Event(); ev.record(s1)
        s2.wait_event(ev)
        with torch.cuda.stream(s1):
            launch(a, 1)
        with torch.cuda.stream(s2):
            launch(b, 2)
        ev2 = torch.cuda.Event(); ev2.record(s2)
        s1.wait_event(ev2)
    for _ in range(5):
        it()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s1)
    for _ in range(n):
        it()
    e1.record(s1); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


args = [int(x) for x in sys.argv[1:]] or [3072, 1024]
for B1, B2 in zip(args[0::2], args[1::2]):
    B = B1 + B2
    print('B = %5d: packed alone %6.1f us | row3 alone %6.1f us | packed %d + row3 %d side by side (two streams, event paired) %6.1f us'
          % (B, alone(B, 1), alone(B, 2), B1, B2, pair(B1, B2)), flush=True)
lib.mfg_set_core_mapping(0)
