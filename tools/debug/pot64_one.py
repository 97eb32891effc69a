import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mjhmc_amd import engine, _lib
from helpers import ref_init_weights
D, N = int(sys.argv[1]), int(sys.argv[2])
W, lognu = ref_init_weights(D, D)
W = W + np.eye(D)
params = np.concatenate([[float(D)], W.ravel(), np.exp(lognu), np.zeros(D)])
ctx = engine.context(0)
en = engine.DeviceEnergy(ctx, _lib.E_PRODUCT_OF_T, D, params)
X0 = np.random.RandomState(5).randn(D, N)
for dtype in ('float32', 'float64'):
    s = engine.DeviceSampler(en, X0, seed=17, dtype=dtype)
    s.set_hparams(0.1, 6, 0.05, 1.0)
    for it in range(3):
        t0 = time.time()
        st, done = s.iterate(1)
        print(dtype, it, 'done', done, [(t.l, t.f, t.r, t.n_cold, t.n_flf_run, t.nonfinite) for t in st], 'EX', s.read(_lib.F_EX)[:3], '%.2fs' % (time.time() - t0), flush=True)
    s.close()
