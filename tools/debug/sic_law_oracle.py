"""float64 MJHMC (the oracle) on the SparseImageCode posterior: does the jump process itself run hot, or only the bf16 chain?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import mjhmc_oracle as orc
from helpers import sic_problem, to_bf16

B, imgs, a0 = sic_problem(0)
y = imgs[:, 0]
N, D = int(sys.argv[2]) if len(sys.argv) > 2 else 256, 1024
variant = sys.argv[1] if len(sys.argv) > 1 else 'f64'


class Fast(orc.Energy):
    def __init__(self, rnd=None):
        orc.Energy.__init__(self)
        self.rnd = rnd or (lambda a: a)
        self.B = self.rnd(B)
    def E_val(self, X):
        R = self.B.dot(self.rnd(X)) - y[:, None]
        return (0.5 * np.sum(R ** 2, axis=0) + 0.01 * np.sum(np.log(1 + X ** 2), axis=0)).reshape((1, -1))
    def dEdX_val(self, X):
        R = self.B.dot(self.rnd(X)) - y[:, None]
        return self.B.T.dot(self.rnd(R)) + 0.01 * 2 * X / (1 + X ** 2)

X0 = a0[:, None] + 0.1 * np.random.RandomState(12).randn(D, N)
np.random.seed(5)
kw = {}
if variant == 'bf16':
    en = Fast(to_bf16); kw['state_rounding'] = to_bf16; X0 = to_bf16(X0)
elif variant == 'bf16state':
    en = Fast(); kw['state_rounding'] = to_bf16; X0 = to_bf16(X0)
elif variant == 'bf16ops':
    en = Fast(to_bf16)
else:
    en = Fast()
cls = orc.ControlHMC if variant == 'control' else orc.MarkovJumpHMC
if variant == 'control':
    s = cls(en, X0, epsilon=0.05, beta=0.1, num_leapfrog_steps=25)
else:
    s = cls(en, X0, epsilon=0.05, beta=0.1, num_leapfrog_steps=25, resample=False, **kw)
t0 = time.time()
for it in range(1, 1501):
    s.sampling_iteration()
    if it in (100, 300, 600, 1000, 1500):
        X, V = s.state.X, s.state.V
        r = y[:, None] - B.dot(X)
        print('%s N %d it %d  e_data %.1f  e_kin %.1f  |a| rms %.2f  l/f/r %.3f/%.3f/%.3f  %.0fs' % (
            variant, N, it, 0.5 * np.mean(np.sum(r ** 2, axis=0)), 0.5 * np.mean(np.sum(V ** 2, axis=0)), np.sqrt(np.mean(X ** 2)),
            s.l_count / float(it * N), s.f_count / float(it * N), s.r_count / float(it * N), time.time() - t0), flush=True)
