"""Calibration: achievable HBM bandwidth on this box (torch device copy / add), for context."""
import torch, time
for mb in (256, 820, 1640):
    n = mb * 1024 * 1024 // 8
    x = torch.randn(n, dtype=torch.float64, device='cuda')
    y = torch.empty_like(x)
    for name, fn in (('copy', lambda: y.copy_(x)), ('scale', lambda: torch.mul(x, 1.0000001, out=y))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print('%s %5d MB: %.4f ms  %.0f GB/s (read+write)' % (name, mb, ms, 2 * n * 8 / ms / 1e6))
