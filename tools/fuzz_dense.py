"""Random shapes and call patterns through the ProductOfT tile kernels: the product library's schedule (F-movers' hand-over,
free-running parts, inverse-L tiles inside the jump launch) against the test build told to integrate every cold cache's
inverse-L proposal (MJHMC_NO_FSPEC) on one stream (MJHMC_NO_SPLIT) -- state, cache, dwelling times, transitions and the
reference's counters bit for bit.   usage: python tools/fuzz_dense.py [seconds, default 60] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mjhmc_amd import engine, _lib                     # noqa: E402
from tests.helpers import hooks_context, ref_init_weights   # noqa: E402

FIELDS = ('X', 'V', 'EX', 'EV', 'HFLF', 'CACHE', 'DWELL', 'TRANS')


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctxs = (engine.context(0), hooks_context(0))
    t0, cases = time.time(), 0
    while time.time() - t0 < budget:
        D = int(rs.choice([8, 36, 64, 100, 128, int(rs.randint(4, 129))]))
        N = int(rs.choice([1, 31, 32, 33, 500, int(rs.randint(1, 4000)), int(rs.randint(8000, 30000))]))
        dtype = 'float64' if rs.rand() < 0.5 else 'float32'
        L = int(rs.choice([1, 2, 4]))
        eps = float(rs.choice([0.05, 0.1]))
        beta = float(rs.choice([0.05, 0.3]))
        W, lognu = ref_init_weights(D, D)
        params = np.concatenate([[float(D)], W.ravel(), np.exp(lognu), np.zeros(D)])
        ens = [engine.DeviceEnergy(c, _lib.E_PRODUCT_OF_T, D, params) for c in ctxs]
        X0 = rs.randn(D, N)
        pair = [engine.DeviceSampler(en, X0, seed=8, dtype=dtype) for en in ens]
        tag = dict(D=D, N=N, dtype=dtype, L=L, eps=eps, beta=beta)
        stats = [[], []]
        for n_it in [int(v) for v in rs.choice([1, 2, 3, 5], size=3)]:
            for k, s in enumerate(pair):
                s.set_hparams(eps, L, -np.log(1 - beta) * 0.5, 1.0)
                if k == 1:
                    os.environ['MJHMC_NO_FSPEC'] = '1'
                    os.environ['MJHMC_NO_SPLIT'] = '1'
                st, done = s.iterate(n_it)
                os.environ.pop('MJHMC_NO_FSPEC', None)
                os.environ.pop('MJHMC_NO_SPLIT', None)
                assert done == n_it, (tag, done)
                stats[k] += list(st)
            for f in FIELDS:
                fa, fb = pair[0].read(getattr(_lib, 'F_' + f)), pair[1].read(getattr(_lib, 'F_' + f))
                assert np.array_equal(fa, fb, equal_nan=True), (tag, len(stats[0]), f)
        for x, y in zip(stats[0], stats[1]):
            assert (x.l, x.f, x.r, x.n_cold, x.E_evals, x.dEdX_evals) == (y.l, y.f, y.r, y.n_cold, y.E_evals, y.dEdX_evals), tag
        for s in pair:
            s.close()
        cases += 1
    print('fuzz_dense: %d random ProductOfT cases in %.0f s, all bit-identical' % (cases, time.time() - t0))


if __name__ == '__main__':
    main()
