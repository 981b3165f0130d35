"""Developer probe: does hiding the batch sums of the main part behind the rollout's last, nearly empty round pay?
Splits the headline batch into 7 exact rounds of tiles + the remainder, runs the remainder's rollout on the caller's
stream while a helper stream sums the main part (fork / join through events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from discrete_mean_field_game_amd import ops

dev = torch.device('cuda:0')
d, T, B = 21, 15, int(sys.argv[1]) if len(sys.argv) > 1 else 65536
slots = 256 * 3
tiles = (B + 11) // 12
main_tiles = (tiles // slots) * slots
B1 = main_tiles * 12
th = torch.tensor([8.86349], dtype=torch.float64, device=dev)
rs = np.random.RandomState(0)
pi0 = torch.as_tensor(rs.dirichlet(np.ones(d), size=B).astype(np.float32), device=dev)
F = ops.num_features(d)
w = torch.as_tensor(rs.rand(F), device=dev)
G = torch.zeros(F + 3, dtype=torch.float64, device=dev)
G2 = torch.zeros(F + 3, dtype=torch.float64, device=dev)
ws = ops.workspace(B * T, d, dev)
ws2 = ops.workspace(B * T, d, dev)
out = ops.rollout(pi0, T, th, 0.16, 12000.0, w=w, seed=1, td=True)
outa = ops.rollout(pi0[:B1], T, th, 0.16, 12000.0, w=w, seed=1, td=True)
outb = ops.rollout(pi0[B1:], T, th, 0.16, 12000.0, w=w, seed=1, td=True, traj_offset=B1)
S = torch.cuda.current_stream()
H = torch.cuda.Stream()

def single():
    ops.rollout(pi0, T, th, 0.16, 12000.0, w=w, seed=1, td=True, out=out)
    ops.grad_accumulate(out['pi_traj'], out['delta'].view(-1), out['g'].view(-1), out['reward'].view(-1), G, ws, T=T)

def split():
    ops.rollout(pi0[:B1], T, th, 0.16, 12000.0, w=w, seed=1, td=True, out=outa)
    e1 = torch.cuda.Event(); e1.record(S)
    H.wait_event(e1)
    with torch.cuda.stream(H):
        ops.grad_accumulate(outa['pi_traj'], outa['delta'].view(-1), outa['g'].view(-1), outa['reward'].view(-1), G2, ws2, T=T)
        e2 = torch.cuda.Event(); e2.record(H)
    ops.rollout(pi0[B1:], T, th, 0.16, 12000.0, w=w, seed=1, td=True, traj_offset=B1, out=outb)
    S.wait_event(e2)
    ops.grad_accumulate(outb['pi_traj'], outb['delta'].view(-1), outb['g'].view(-1), outb['reward'].view(-1), G2, ws, T=T, accumulate=True)

def serial_split():
    ops.rollout(pi0[:B1], T, th, 0.16, 12000.0, w=w, seed=1, td=True, out=outa)
    ops.rollout(pi0[B1:], T, th, 0.16, 12000.0, w=w, seed=1, td=True, traj_offset=B1, out=outb)

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best * 1e3

print('B=%d tiles=%d main=%d (%d trajectories) tail=%d trajectories' % (B, tiles, main_tiles, B1, B - B1))
print('single rollout + sums            %8.1f us' % t(single))
print('main | tail rollout, no sums     %8.1f us' % t(serial_split))
print('main, then tail rollout || sums  %8.1f us' % t(split))
single(); split(); torch.cuda.synchronize()
print('G agreement: max rel diff %.2e' % float(((G - G2).abs().max() / G.abs().max())))
print('trajectories identical: %s' % bool(torch.equal(out['pi_traj'][:B1], outa['pi_traj']) and torch.equal(out['pi_traj'][B1:], outb['pi_traj'])))
