"""Per-iteration device time of the three sampler modes on the C2 shape."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib          # noqa: E402

D, N = (int(a) for a in sys.argv[1:3]) if len(sys.argv) >= 3 else (512, 100000)
ctx = engine.context(0)
X0 = np.random.RandomState(0).randn(D, N)
en = engine.DeviceEnergy(ctx, _lib.E_ISO_GAUSS, D, [1.0])
for name, mode in (('MJHMC', _lib.MODE_MJHMC), ('ControlHMC', _lib.MODE_CONTROL), ('ContinuousTimeHMC', _lib.MODE_CTHMC)):
    smp = engine.DeviceSampler(en, X0, seed=1, mode=mode)
    smp.set_hparams(0.05, 10, 0.0527, 1.0, 1.0)
    smp.iterate(128)
    best = 1e9
    for _ in range(3):
        smp.iterate(64)
        best = min(best, smp.last_timing()['total_ms'] / 64)
    print('%-18s %dx%d  %.4f ms per iteration' % (name, D, N, best), flush=True)
    smp.close()
