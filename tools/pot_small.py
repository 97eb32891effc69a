import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mjhmc_amd import engine, _lib
ctx = engine.context(0)
for D, N in ((36, 100), (36, 1000), (100, 1000), (512, 1000)):
    W, lognu = bench.pot_model(D)
    params = np.concatenate([[float(D)], W.ravel(), np.exp(lognu), np.zeros(D)])
    en = engine.DeviceEnergy(ctx, _lib.E_PRODUCT_OF_T, D, params)
    X0 = np.random.RandomState(0).randn(D, N)
    for dtype in ('float32', 'float64'):
        smp = engine.DeviceSampler(en, X0, seed=1, dtype=dtype)
        smp.set_hparams(0.05, 10, 0.0527, 1.0, 0.5)
        smp.iterate(64)
        t0 = time.perf_counter(); smp.iterate(256); smp.sync(); wall = (time.perf_counter() - t0) / 256 * 1e6
        print('PoT D=%3d N=%5d %s state: %.1f us/iter wall, %.1f us device' % (D, N, dtype, wall, smp.last_timing()['total_ms'] / 256 * 1e3), flush=True)
        smp.close()
