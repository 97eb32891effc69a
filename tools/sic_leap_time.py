"""Kernel time of the SparseImageCode trajectory alone (mjhmc_leapfrog on caller states: L leapfrog steps, no jump logic):
run under `rocprofv3 --kernel-trace --stats` and read sic_leap_kernel's duration.  MJHMC_HIP_LIB selects the library
(tools/sic_variants.sh builds timing variants).  usage: python tools/sic_leap_time.py [tiles_per_cu] [L]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib          # noqa: E402
from bench import sic_model                 # noqa: E402

tiles_per_cu = int(sys.argv[1]) if len(sys.argv) > 1 else 4
L = int(sys.argv[2]) if len(sys.argv) > 2 else 25
ctx = engine.context(0)
ncu = ctx.info()['n_cu']
N = 32 * ncu * tiles_per_cu
B, y, a0 = sic_model()
en = engine.DeviceEnergy(ctx, _lib.E_SPARSE_CODE, 1024, np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), y]))
rs = np.random.RandomState(0)
X = a0[:, None] + 0.1 * rs.randn(1024, N)
V = rs.randn(1024, N)
for rep in range(3):
    t0 = time.perf_counter()
    en.leapfrog(X, V, 0.05, L, want_grad=False, dtype='bfloat16')
    dt = time.perf_counter() - t0
print('N', N, 'L', L, 'wall of the last call (with transfers) %.3f s' % dt, 'lib', _lib.LIB_PATH)
