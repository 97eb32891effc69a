import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mjhmc_amd import engine, _lib
from helpers import hooks_context, ref_init_weights
what = sys.argv[1] if len(sys.argv) > 1 else 'pot36'
D, N = 36, int(sys.argv[2]) if len(sys.argv) > 2 else 20000
dtype = 'float64' if what == 'pot36f64' else 'float32'
W, lognu = ref_init_weights(D, D)
params = np.concatenate([[float(D)], W.ravel(), np.exp(lognu), np.zeros(D)])
ctxs = (engine.context(0), hooks_context(0))
ens = [engine.DeviceEnergy(c, _lib.E_PRODUCT_OF_T, D, params) for c in ctxs]
X0 = np.random.RandomState(3).randn(D, N)
pair = [engine.DeviceSampler(en, X0, seed=8, dtype=dtype) for en in ens]
hist = []
for it in range(6):
    for k, s in enumerate(pair):
        s.set_hparams(0.1, 6, 0.1, 1.0)
        if k == 1: os.environ['MJHMC_NO_FSPEC'] = '1'
        else: os.environ.pop('MJHMC_NO_FSPEC', None)
        st, done = s.iterate(1)
        print(it, k, [(t.l, t.f, t.r, t.n_cold, t.n_flf_run) for t in st])
    os.environ.pop('MJHMC_NO_FSPEC', None)
    ta, tb = pair[0].read(_lib.F_TRANS), pair[1].read(_lib.F_TRANS)
    hist.append(ta.copy())
    for f in ('X', 'V', 'EX', 'EV', 'HFLF', 'DWELL', 'TRANS'):
        fa, fb = pair[0].read(getattr(_lib, 'F_' + f)), pair[1].read(getattr(_lib, 'F_' + f))
        if not np.array_equal(fa, fb, equal_nan=True):
            bad = np.where(~((fa == fb) | (np.isnan(fa) & np.isnan(fb))))
            cols = np.unique(bad[-1])
            print('  it', it, f, 'differs in', len(cols), 'particles; first', cols[:8], 'prev trans', [h[cols[:8]] for h in hist])
            if f == 'DWELL':
                print('   dwell a', fa[cols[:4]], 'b', fb[cols[:4]])
