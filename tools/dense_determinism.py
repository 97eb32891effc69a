"""Run-to-run determinism of the dense kernels at full size: the same sampler twice from the same state and seed, N
iterations each, states compared BITWISE after every call.  The SparseImageCode rounds hand LDS images from wave to wave
through barriers, vmcnt waits and LDS-DMA requests made by OTHER waves (dense_sic.hip: issue_partner); a race there would
show up as a run-to-run difference long before a parity test against the oracle catches it.
usage: python tools/dense_determinism.py [c5|c3] [calls] [iterations per call] [nparticles]"""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib              # noqa: E402
import bench                                    # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else 'c5'
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
per_call = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w = dict(bench.WORKLOADS[key])
n = int(sys.argv[4]) if len(sys.argv) > 4 else w['N']
ctx = engine.context(0)
kind = {'sic': _lib.E_SPARSE_CODE, 'pot': _lib.E_PRODUCT_OF_T}[w['kind']]
if w['kind'] == 'sic':
    B, y, _ = bench.sic_model()
    params = np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), y])
else:
    W, lognu = bench.pot_model(w['D'])
    params = np.concatenate([[float(w['D'])], W.ravel(), np.exp(lognu), np.zeros(w['D'])])
en = engine.DeviceEnergy(ctx, kind, w['D'], params)
X0 = bench.initial_state(w, n, 0)
p_r = -np.log(1 - w['beta']) * 0.5
crcs = []
for run in range(2):
    smp = engine.DeviceSampler(en, X0, seed=4242, first_particle_id=0, dtype=w['dtype'])
    smp.set_hparams(w['eps'], w['L'], p_r, 1.0)
    seq = []
    for c in range(calls):
        stats, done = smp.iterate(per_call)
        assert done == per_call
        X = smp.read(_lib.F_X)
        V = smp.read(_lib.F_V)
        seq.append((zlib.crc32(X.tobytes()), zlib.crc32(V.tobytes()), sum(s.l for s in stats), sum(s.r for s in stats)))
    crcs.append(seq)
    del smp
bad = [i for i in range(calls) if crcs[0][i] != crcs[1][i]]
print(key, 'N', n, 'calls', calls, 'x', per_call, 'iterations:', 'IDENTICAL' if not bad else 'DIFFER at calls %s' % bad[:10])
print('last call (crc X, crc V, L moves, R moves):', crcs[0][-1])
sys.exit(1 if bad else 0)
