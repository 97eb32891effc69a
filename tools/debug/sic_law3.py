import os, sys
import numpy as np
from scipy import stats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mjhmc_amd import engine, _lib
from helpers import sic_problem, to_bf16
ctx = engine.context(0)
B, imgs, a0 = sic_problem(0)
y = imgs[:, 0]
Bq = to_bf16(B)
N, D = 4096, 1024
en = engine.DeviceEnergy(ctx, _lib.E_SPARSE_CODE, D, np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), y]))
X0 = a0[:, None] + 0.1 * np.random.RandomState(12).randn(D, N)
for mode_name, state in (('CONTROL', 'bfloat16'), ('CONTROL', 'float32'), ('MJHMC', 'float32'), ('MJHMC', 'bfloat16')):
    for seed in (2027, 5):
        mode = {'MJHMC': _lib.MODE_MJHMC, 'CONTROL': _lib.MODE_CONTROL}[mode_name]
        s = engine.DeviceSampler(en, X0, seed=seed, dtype=state, mode=mode)
        s.set_hparams(0.05, 25, 0.0527 if mode_name == 'MJHMC' else 0.3, 0.1 if mode_name == 'MJHMC' else 1.0, 1.0)
        out = []
        for tgt in (500, 1500, 3000):
            while len(out) < 0: pass
            n = tgt - (0 if not out else out[-1][0])
            for _ in range(n // 100): s.iterate(100)
            X, V = s.read(_lib.F_X), s.read(_lib.F_V)
            r = y[:, None] - B.dot(X)
            rq = y[:, None] - Bq.dot(to_bf16(X))
            e, eq, ek = 0.5 * np.sum(r ** 2, axis=0), 0.5 * np.sum(rq ** 2, axis=0), 0.5 * np.sum(V ** 2, axis=0)
            out.append((tgt, e.mean(), eq.mean(), ek.mean(), stats.kstest(e, 'gamma', args=(128.0,)).pvalue, stats.kstest(eq, 'gamma', args=(128.0,)).pvalue,
                        np.sqrt(np.mean(X ** 2))))
        print(mode_name, state, seed, ' | '.join('it %d e %.2f e_rounded_operands %.2f ekin %.2f p %.1e p_q %.1e rms %.1f' % o for o in out), flush=True)
        s.close()
