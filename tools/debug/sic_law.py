import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mjhmc_amd import engine, _lib
from helpers import sic_problem
ctx = engine.context(0)
B, imgs, a0 = sic_problem(0)
y = imgs[:, 0]
N, D = 4096, 1024
en = engine.DeviceEnergy(ctx, _lib.E_SPARSE_CODE, D, np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), y]))
X0 = a0[:, None] + 0.1 * np.random.RandomState(12).randn(D, N)
for mode_name, eps, L in (('MJHMC', 0.05, 25), ('CONTROL', 0.05, 25), ('MJHMC', 0.0125, 100), ('CONTROL', 0.0125, 100), ('MJHMC', 0.2, 6)):
    mode = {'MJHMC': _lib.MODE_MJHMC, 'CONTROL': _lib.MODE_CONTROL}[mode_name]
    s = engine.DeviceSampler(en, X0, seed=2027, dtype='bfloat16', mode=mode)
    s.set_hparams(eps, L, 0.0527 if mode_name == 'MJHMC' else 0.3, 0.1 if mode_name == 'MJHMC' else 1.0, 1.0)
    done_total = 0
    for tgt in (100, 300, 1000, 2000):
        mv = np.zeros(4)
        while done_total < tgt:
            st, done = s.iterate(100)
            done_total += done
            mv += [sum(t.l for t in st), sum(t.f for t in st), sum(t.r for t in st), sum(t.fl for t in st)]
        X, V = s.read(_lib.F_X), s.read(_lib.F_V)
        EXd, EVd = s.read(_lib.F_EX), s.read(_lib.F_EV)
        r = y[:, None] - B.dot(X)
        e_data, e_kin = 0.5 * np.sum(r ** 2, axis=0), 0.5 * np.sum(V ** 2, axis=0)
        prior = 0.01 * np.sum(np.log1p(X ** 2), axis=0)
        print('%-8s eps %.4f L %3d it %5d  e_data %.1f (want 128)  e_kin %.1f (want 512)  prior %.2f  dev EX-host %.3f  dev EV-host %.3f  |a| rms %.3f  moves %s'
              % (mode_name, eps, L, done_total, e_data.mean(), e_kin.mean(), prior.mean(), np.mean(EXd - e_data - prior), np.mean(EVd - e_kin),
                 np.sqrt(np.mean(X ** 2)), (mv / mv.sum()).round(3).tolist()), flush=True)
    s.close()
