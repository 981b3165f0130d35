"""Given-P kernel (transition + reward) on a > L3 slab, event timed (developer tool).  usage: step_probe.py [d,N ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from discrete_mean_field_game_amd import ops
dev = torch.device('cuda:0')

def probe(d, N, want_reward=True):
    g = torch.Generator(device=dev); g.manual_seed(0)
    pi = torch.rand(N, d, device=dev, generator=g); pi = (pi / pi.sum(1, keepdim=True)).contiguous()
    P = torch.rand(N, d, d, device=dev, generator=g)
    P /= P.sum(-1, keepdim=True)
    for _ in range(20): ops.step_given_P(pi, P, want_reward=want_reward)
    torch.cuda.synchronize()
    best = 1e9; tot = 0.0
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.step_given_P(pi, P, want_reward=want_reward)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e-3
        best = min(best, t); tot += t
    bps = 4 * (d * d + 2 * d + 1)
    print(('' if want_reward else '[no reward] ') + 'd=%d N=%d slab %.2f GB: avg %.1f us (best %.1f)  %.2f TB/s = %.1f%% of 8 TB/s' % (
        d, N, N * d * d * 4 / 1e9, tot / 3 * 1e6, best * 1e6, N * bps / (tot / 3) / 1e12, 100 * N * bps / (tot / 3) / 8e12), flush=True)

if __name__ == '__main__':
    cfgs = [(21, 983040), (15, 1966080), (128, 16384), (256, 16384)]
    if len(sys.argv) > 1:
        cfgs = [tuple(int(x) for x in a.split(',')) for a in sys.argv[1:]]
    for c in cfgs:
        probe(*c)
        if os.environ.get('STEP_PROBE_NOREWARD'):
            probe(*c, want_reward=False)
