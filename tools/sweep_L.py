"""Perf exploration: jump-kernel time vs number of leapfrog steps (is the kernel HBM- or VALU-bound?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mjhmc_amd import engine, _lib

D, N = int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 100000
kind = sys.argv[3] if len(sys.argv) > 3 else 'iso'
ctx = engine.context(0)
en = engine.DeviceEnergy(ctx, _lib.E_ISO_GAUSS if kind == 'iso' else _lib.E_FUNNEL_NEAL, D, [1.0 if kind == 'iso' else 3.0])
X0 = np.random.RandomState(0).randn(D, N)
s = engine.DeviceSampler(en, X0, seed=1)
for L in (0, 1, 2, 5, 10, 20, 40):
    s.set_hparams(0.05 if L else 0.0, L, 0.05, 1.0)
    s.iterate(3)
    st, done = s.iterate(20)
    t = s.last_timing()
    ms = t['jump_kernel_ms'] / t['n_jump_launches']
    cold = sum(x.n_cold for x in st) / (20.0 * N)
    print('L=%3d  kernel %.4f ms  total/iter %.4f ms  cold %.3f  alg GB/s %.0f' % (L, ms, t['total_ms'] / 20, cold, (4 * D * 8 + 59) * N / ms / 1e6))
