"""2000 single-iteration calls on a tiny batch (for an API trace: where a sampling_iteration() call's time goes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mjhmc_amd import engine, _lib
ctx = engine.context(0)
smp = engine.DeviceSampler(engine.DeviceEnergy(ctx, _lib.E_ISO_GAUSS, 2, [1.0]), np.random.RandomState(0).randn(2, 100), seed=1)
smp.set_hparams(0.05, 10, 0.05, 1.0, 0.5)
for _ in range(200):
    smp.iterate(1)
t0 = time.perf_counter()
for _ in range(2000):
    smp.iterate(1)
smp.sync()
print('%.1f us per call' % ((time.perf_counter() - t0) / 2000 * 1e6))
