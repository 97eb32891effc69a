"""Random shapes through the row-form kernels of the funnels: a fused call (mjhmc_fused_rows_relay_kernel, ring snapshots included)
against the same iterations one call at a time (below 16 384 particles: the jump kernel, a group of lanes per particle;
above: the trajectory launch in row form + the jump-process launch) -- state, scalars, ring and counters bit for bit.
With the test build (MJHMC_HIP_LIB=.../libmjhmc_hip_test.so) and MJHMC_FORCE_RELAY=1 the relay kernel takes the short
trajectories too (parts of one leapfrog step, empty parts).
usage: python tools/fuzz_rows.py [seconds, default 60] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib          # noqa: E402

FIELDS = ('X', 'V', 'EX', 'EV', 'HFLF', 'DWELL', 'TRANS')


def same_bits(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = engine.context(0)
    t0, cases = time.time(), 0
    while time.time() - t0 < budget:
        D = int(rs.choice([9, 10, 15, 16, 17, 21, 31, 32, int(rs.randint(9, 33))]))
        N = int(rs.choice([1, 63, 64, 65, 127, 128, 1000, int(rs.randint(1, 5000)), int(rs.randint(16384, 40000))]))
        u = rs.rand()
        kind = _lib.E_FUNNEL_NEAL if u < 0.6 else (_lib.E_FUNNEL_REF if u < 0.75 else _lib.E_MM_GAUSS)   # the energies with a row form
        n_iter = int(rs.choice([2, 3, 7, 20, 64, 65, 70]))
        if N > 10000:
            n_iter = min(n_iter, 7)
        L = int(rs.choice([0, 1, 2, 5, 9, 12, 13, 16, 23]))     # (from 12 steps up the fused call is the relay kernel's)
        eps = float(rs.choice([0.01, 0.05, 0.1]))
        with_ring = rs.rand() < 0.5
        X0 = rs.randn(D, N) * 0.7
        en_a = engine.DeviceEnergy(ctx, kind, D, [3.0])
        en_b = engine.DeviceEnergy(ctx, kind, D, [3.0])
        a = engine.DeviceSampler(en_a, X0, seed=5)
        b = engine.DeviceSampler(en_b, X0, seed=5)
        for s in (a, b):
            s.set_hparams(eps, L, 0.1, 1.0, 0.5)
            if with_ring:
                s.ring_alloc(n_iter)
        sa, da = a.iterate(n_iter, ring_slot0=0) if with_ring else a.iterate(n_iter)
        sb, db = [], 0
        for i in range(n_iter):
            st, d = b.iterate(1, ring_slot0=i) if with_ring else b.iterate(1)
            sb.append(st[0])
            db += d
            if d == 0:
                break
        tag = dict(D=D, N=N, kind=kind, n_iter=n_iter, L=L, eps=eps, ring=with_ring)
        assert da == db, (tag, da, db)
        fields = FIELDS if da == n_iter else ('X', 'V', 'EX', 'EV', 'HFLF')
        for f in fields:
            assert same_bits(a.read(getattr(_lib, 'F_' + f)), b.read(getattr(_lib, 'F_' + f))), (tag, f)
        for x, y in zip(sa[:da], sb[:da]):
            assert (x.l, x.f, x.r, x.n_cold, x.E_evals, x.dEdX_evals) == (y.l, y.f, y.r, y.n_cold, y.E_evals, y.dEdX_evals), tag
        if with_ring and da > 0:
            assert same_bits(a.ring_read(0, da, stacked=True), b.ring_read(0, da, stacked=True)), (tag, 'ring')
            assert same_bits(a.ring_read_dwell(0, da), b.ring_read_dwell(0, da)), (tag, 'dwell ring')
        a.close()
        b.close()
        cases += 1
    print('fuzz_rows: %d random cases in %.0f s, all bit-identical' % (cases, time.time() - t0))


if __name__ == '__main__':
    main()
