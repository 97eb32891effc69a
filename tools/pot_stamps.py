"""Reads the cycle stamps of the ProductOfT timing build (tools/pot_stamps.sh): the parts of the LAST gradient evaluation
workgroup 0 ran, per wave.  usage: MJHMC_HIP_LIB=.../libpot_stamps.so python tools/pot_stamps.py [nparticles]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib              # noqa: E402
import bench                                    # noqa: E402

w = dict(bench.WORKLOADS['c3'])
n = int(sys.argv[1]) if len(sys.argv) > 1 else w['N']
ctx = engine.context(0)
W, lognu = bench.pot_model(w['D'])
params = np.concatenate([[float(w['D'])], W.ravel(), np.exp(lognu), np.zeros(w['D'])])
en = engine.DeviceEnergy(ctx, _lib.E_PRODUCT_OF_T, w['D'], params)
smp = engine.DeviceSampler(en, bench.initial_state(w, n, 0), seed=1, first_particle_id=0, dtype=w['dtype'])
smp.set_hparams(w['eps'], w['L'], -np.log(1 - w['beta']) * 0.5, 1.0)
smp.iterate(3)
smp.sync()
lib = _lib.load()
st = np.zeros((4, 8, 8), dtype=np.uint64)     # [wave][slot][part]; the float32-state kernels write slot 0
rc = lib.mjhmc_pot_stamps(st.ctypes.data_as(ctypes.c_void_p))
st = st.astype(np.int64)[:, 0, :]
names = ['enter', 'past barrier 1', 'GEMM1 done', 'phi + publish done', 'past barrier 2', 'GEMM2 done']
print('rc', rc, ' cycles since the wave entered the gradient (MFMA time of one GEMM: 1024 x 64 = 65 536)')
for wv in range(4):
    print('wave', wv, ' '.join('%s=%d' % (names[i], st[wv, i] - st[wv, 0]) for i in range(6)))
