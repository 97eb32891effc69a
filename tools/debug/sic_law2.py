import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mjhmc_amd import engine, _lib
from helpers import sic_problem
ctx = engine.context(0)
B, imgs, a0 = sic_problem(0)
y = imgs[:, 0]
N, D = 2048, 1024
en = engine.DeviceEnergy(ctx, _lib.E_SPARSE_CODE, D, np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), y]))
X0 = a0[:, None] + 0.1 * np.random.RandomState(12).randn(D, N)
for burn in (300, 1500):
    s = engine.DeviceSampler(en, X0, seed=2027, dtype='bfloat16', mode=_lib.MODE_MJHMC)
    s.set_hparams(0.05, 25, 0.0527, 0.1, 1.0)
    for _ in range(burn // 100):
        s.iterate(100)
    w_sum = 0.0; acc = np.zeros(2); raw = np.zeros(2); n = 0
    dw_all = []
    for t in range(120):
        X, V = s.read(_lib.F_X), s.read(_lib.F_V)
        r = y[:, None] - B.dot(X)
        e_data, e_kin = 0.5 * np.sum(r ** 2, axis=0), 0.5 * np.sum(V ** 2, axis=0)
        s.iterate(1)
        dw = s.read(_lib.F_DWELL)
        w_sum += dw.sum(); acc += [np.sum(dw * e_data), np.sum(dw * e_kin)]
        raw += [e_data.mean(), e_kin.mean()]; n += 1
        dw_all.append(dw)
    dw_all = np.concatenate(dw_all)
    print('burn %d: dwell-weighted e_data %.1f e_kin %.1f | unweighted e_data %.1f e_kin %.1f | dwell mean %.3f std %.3f min %.2e max %.2f'
          % (burn, acc[0] / w_sum, acc[1] / w_sum, raw[0] / n, raw[1] / n, dw_all.mean(), dw_all.std(), dw_all.min(), dw_all.max()), flush=True)
    s.close()
