"""Long runs of the five bench workloads (thousands of sampling iterations each): every iteration's tallies must add up
to the batch, the run must never meet a non-finite rate, and the energies stored with the final state must equal a fresh
evaluation of that state.  Exercises the multi-launch paths over many calls: fused launches (C2), the compacted passes
with the refresh riding in the inverse-L pass (C4), the two-stream halves of the dense energies (C3, C5).
usage: python tools/soak_all.py [seconds per workload, default 40]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mjhmc_amd import engine, _lib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
ctx = engine.context(0)
for key in ('c2', 'c4', 'c3', 'c3f64', 'c5', 'c5bf16'):
    w = dict(bench.WORKLOADS[key])
    kind = {'iso': _lib.E_ISO_GAUSS, 'funnel': _lib.E_FUNNEL_NEAL, 'pot': _lib.E_PRODUCT_OF_T, 'sic': _lib.E_SPARSE_CODE}[w['kind']]
    params = w['params']
    if w['kind'] == 'pot':
        W, lognu = bench.pot_model(w['D'])
        params = np.concatenate([[float(w['D'])], W.ravel(), np.exp(lognu), np.zeros(w['D'])])
    if w['kind'] == 'sic':
        B, y, _ = bench.sic_model()
        params = np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), y])
    en = engine.DeviceEnergy(ctx, kind, w['D'], params)
    N = w['N']
    s = engine.DeviceSampler(en, bench.initial_state(w, N, 0), seed=99, dtype=w['dtype'])
    s.set_hparams(w['eps'], w['L'], -np.log(1 - w['beta']) * 0.5, 1.0)
    per_call = 64 if key in ('c2', 'c4') else 8
    tot = np.zeros(3, dtype=np.int64)
    n_it, t0 = 0, time.time()
    while time.time() - t0 < budget:
        st, done = s.iterate(per_call)
        assert done == per_call, (key, n_it, 'a non-finite rate interrupted the run')
        for x in st:
            assert x.l + x.f + x.r == N, (key, n_it, x.l, x.f, x.r)
            assert x.E_evals == N + x.n_cold and x.dEdX_evals == w['L'] * (N + x.n_cold), (key, n_it)
            tot += (x.l, x.f, x.r)
        n_it += per_call
    # stored energies == a fresh evaluation of the stored state (a column subset: the dense evaluations are not free)
    cols = np.random.RandomState(1).choice(N, 4096, replace=False)
    X = s.read(_lib.F_X)[:, cols]
    EX = s.read(_lib.F_EX)[cols]
    E, _ = en.eval(X, want_E=True, want_grad=False, dtype=w['dtype'])
    tol = {'float64': 1e-11, 'float32': 2e-5, 'bfloat16': 5e-4}[w['dtype']]
    if w['kind'] == 'pot':
        tol = 2e-5          # (float64 state too: the energy is a float32 matrix product in both arithmetics)
    err = np.abs(E - EX).max() / max(1.0, np.abs(EX).max())
    assert err < tol, (key, 'stored EX differs from E(X)', err)
    assert np.isfinite(s.read(_lib.F_V)).all()
    print('%s: %d iterations in %.0f s ok; l/f/r fractions %s; max |EX - E(X)| / max |EX| = %.2e'
          % (key, n_it, time.time() - t0, np.round(tot / tot.sum(), 4), err))
    s.close()
