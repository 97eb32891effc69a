import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from mjhmc_amd import engine, _lib
ctx = engine.context(0)
D, N = 512, 100000
X0 = np.random.RandomState(0).randn(D, N)
s = engine.DeviceSampler(engine.DeviceEnergy(ctx, _lib.E_ISO_GAUSS, D, [1.0]), X0, seed=5)
s.set_hparams(0.05, 10, 0.0527, 1.0, 0.5)
tot = np.zeros(3, dtype=np.int64)
for rep in range(20):
    st, done = s.iterate(500)
    assert done == 500
    for x in st:
        tot += (x.l, x.f, x.r)
        assert x.l + x.f + x.r == N
X, V, EX, EV = s.read(_lib.F_X), s.read(_lib.F_V), s.read(_lib.F_EX), s.read(_lib.F_EV)
assert np.allclose(EX, (X ** 2).sum(0) / 2, rtol=1e-12) and np.allclose(EV, (V ** 2).sum(0) / 2, rtol=1e-12)
print('10000 iterations ok; l/f/r fractions', tot / tot.sum(), ' <x^2> =', (X ** 2).mean(), ' <v^2> =', (V ** 2).mean())
