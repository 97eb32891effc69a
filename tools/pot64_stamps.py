"""Reads the cycle stamps of the timing build of the float64-state ProductOfT kernel (tools/pot_stamps.sh dense_pot64):
per leapfrog step of the LAST trajectory workgroup 0 ran, per wave -- the parts of the gradient and the streamed
kick / drift pass behind it.  usage: MJHMC_HIP_LIB=.../libpot_stamps.so python tools/pot64_stamps.py [nparticles]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib              # noqa: E402
import bench                                    # noqa: E402

w = dict(bench.WORKLOADS['c3'])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192      # one tile per CU, one stream
ctx = engine.context(0)
W, lognu = bench.pot_model(w['D'])
params = np.concatenate([[float(w['D'])], W.ravel(), np.exp(lognu), np.zeros(w['D'])])
en = engine.DeviceEnergy(ctx, _lib.E_PRODUCT_OF_T, w['D'], params)
smp = engine.DeviceSampler(en, bench.initial_state(w, n, 0), seed=1, first_particle_id=0, dtype='float64')
smp.set_hparams(w['eps'], w['L'], -np.log(1 - w['beta']) * 0.5, 1.0)
smp.iterate(3)
smp.sync()
lib = _lib.load()
st = np.zeros((4, 8, 8), dtype=np.uint64)
rc = lib.mjhmc_pot64_stamps(st.ctypes.data_as(ctypes.c_void_p))
st = st.astype(np.int64)
print('rc', rc, ' cycles (MFMA time of one GEMM: 1024 x 64 = 65 536); per step: barrier 1 | GEMM 1 | phi + publish | barrier 2 | '
      'GEMM 2 | kick / drift pass | whole step')
for wv in range(4):
    for s in range(1, 6):
        t = st[wv, s]
        print('wave %d step %d: %6d | %6d | %6d | %6d | %6d | %6d | %7d' %
              (wv, s, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[7] - t[6], t[7] - st[wv, s - 1][7]))
