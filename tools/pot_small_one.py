import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mjhmc_amd import engine, _lib
ctx = engine.context(0)
D, N = 36, 1000
W, lognu = bench.pot_model(D)
params = np.concatenate([[float(D)], W.ravel(), np.exp(lognu), np.zeros(D)])
en = engine.DeviceEnergy(ctx, _lib.E_PRODUCT_OF_T, D, params)
X0 = np.random.RandomState(0).randn(D, N)
smp = engine.DeviceSampler(en, X0, seed=1, dtype=sys.argv[1])
smp.set_hparams(0.05, 10, 0.0527, 1.0, 0.5)
smp.iterate(64)
smp.iterate(256); smp.sync()
