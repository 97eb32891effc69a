"""C4 (Neal funnel 32 x N, L 15, float64): a few compacted iterations, for profiler runs.
usage: python tools/c4_iter.py [N] [iterations] [L] [beta] [funnel|mm]      (test build when MJHMC_HIP_LIB names it: MJHMC_NO_ROWS,
MJHMC_NO_RELAY, MJHMC_FUSE_BELOW; mm: MultimodalGaussian 32 x N instead of the funnel)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib          # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
it = int(sys.argv[2]) if len(sys.argv) > 2 else 20
L = int(sys.argv[3]) if len(sys.argv) > 3 else 15
beta = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1      # the refresh rate p_r = -log(1 - beta) / 2
ctx = engine.context(0)
rng = np.random.RandomState(0)
X0 = rng.randn(32, N)
X0[0] *= 3.0
X0[1:] *= np.exp(X0[0] / 2.)
kind = sys.argv[5] if len(sys.argv) > 5 else 'funnel'
if kind == 'mm':
    X0 = rng.randn(32, N)
    X0[0] += 6.0 * (rng.rand(N) < 0.5) - 3.0
en = engine.DeviceEnergy(ctx, _lib.E_MM_GAUSS if kind == 'mm' else _lib.E_FUNNEL_NEAL, 32, [3.0])
smp = engine.DeviceSampler(en, X0, seed=1)
smp.set_hparams(0.05, L, -np.log(1.0 - beta) * 0.5, 1.0)
if it == 1:      # the sampling_iteration() path: calls of ONE iteration; the mean over 20 warm calls
    tot = 0.0
    for k in range(30):
        smp.iterate(1)
        smp.sync()
        if k >= 10:
            tot += smp.last_timing()['total_ms']
    print('total_ms per one-iteration call', tot / 20)
else:
    for _ in range(3):
        smp.iterate(it)
        smp.sync()
    print('total_ms per iteration', smp.last_timing()['total_ms'] / it)
