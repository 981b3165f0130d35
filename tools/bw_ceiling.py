"""Measure this box's achievable HBM read / copy rates with stock torch kernels (calibration of the roofline)."""
import torch, time
dev = torch.device('cuda:0')
x = torch.rand(400_000_000, device=dev)          # 1.6 GB
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
y = torch.empty_like(x)
t = timeit(lambda: x.sum()); print('read  (sum)   %.2f TB/s' % (x.numel() * 4 / t / 1e12))
t = timeit(lambda: y.copy_(x)); print('copy  (r+w)   %.2f TB/s' % (2 * x.numel() * 4 / t / 1e12))
t = timeit(lambda: y.fill_(1.0)); print('write (fill)  %.2f TB/s' % (x.numel() * 4 / t / 1e12))
t = timeit(lambda: torch.max(x)); print('read  (max)   %.2f TB/s' % (x.numel() * 4 / t / 1e12))
