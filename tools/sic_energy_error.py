"""Leapfrog energy error |H(L z) - H(z)| of the bf16 SparseImageCode kernel through mjhmc_leapfrog, at eps, eps/2, eps/4
with the trajectory length held: the numbers behind tests/test_gpu_stationary.py::test_sic_leapfrog_conserves_energy.
usage (GPU box): python tools/sic_energy_error.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib              # noqa: E402
from tests.helpers import sic_problem, to_bf16  # noqa: E402

ctx = engine.context(0)
for nc in (1024, 512):
    for P in (1, 9):
        for cauchy in (True, False):
            B, imgs, a0 = sic_problem(3, n_patches=P, n_coeffs=nc)
            N = 512
            rs = np.random.RandomState(1)
            X = to_bf16(a0[:, None] + 0.1 * rs.randn(P * nc, N))
            V = to_bf16(rs.randn(P * nc, N))
            for lam, tag in ((0.01, 'right'), (None, 'H of 2 lambda')):
                params = np.concatenate([[float(P), 256.0, float(nc), 0.01, 1.0 if cauchy else 0.0], B.ravel(), imgs[:, :P].T.ravel()])
                en = engine.DeviceEnergy(ctx, _lib.E_SPARSE_CODE, P * nc, params)
                params2 = params.copy()
                params2[3] = 0.02 if lam is None else 0.01
                en_h = engine.DeviceEnergy(ctx, _lib.E_SPARSE_CODE, P * nc, params2)       # the energy H is measured with
                row = []
                for eps, L in ((0.1, 4), (0.05, 8), (0.025, 16), (0.0125, 32)):
                    E0, _ = en_h.eval(X, want_E=True, want_grad=False, dtype='bfloat16')
                    Xo, Vo, EX, EV, _ = en.leapfrog(X, V, eps, L, want_grad=False, dtype='bfloat16')
                    E1, _ = en_h.eval(Xo, want_E=True, want_grad=False, dtype='bfloat16')
                    dH = (E1 + 0.5 * np.sum(Vo ** 2, axis=0)) - (E0 + 0.5 * np.sum(V ** 2, axis=0))
                    row.append((float(np.median(np.abs(dH))), float(np.abs(np.mean(dH)))))
                print('nc %4d P %d %-7s %-14s median |dH| / |mean dH| at eps 0.1, 0.05, 0.025, 0.0125: %s' %
                      (nc, P, 'cauchy' if cauchy else 'laplace', tag, ' '.join('%.4f/%.4f' % r for r in row)))
