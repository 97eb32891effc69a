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

tiles_per_cu = float(sys.argv[1]) if len(sys.argv) > 1 else 4
L = int(sys.argv[2]) if len(sys.argv) > 2 else 25
ctx = engine.context(0)
ncu = ctx.info()['n_cu']
N = int(32 * ncu * tiles_per_cu)
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
lib = _lib.load()
if hasattr(lib, 'mjhmc_sic_stamps'):       # timing build 30: phase stamps of four of workgroup 0's waves in the last fused pass
    import ctypes
    st = np.zeros((4, 8, 8), dtype=np.uint32)
    rc = lib.mjhmc_sic_stamps(st.ctypes.data_as(ctypes.c_void_p))
    st = (st - st[:, 0, 0].min()).astype(np.int64)           # (unsigned wrap-around is harmless)
    if os.environ.get('SIC_STAMP_OUT'):                      # one position per build: tools/sic_stamps_merge.py joins them
        np.save(os.environ['SIC_STAMP_OUT'], st)
    names = ['start', 'G2 done', 'at A', 'past A', 'G1 done', 'past B', 'issued']
    print('stamps rc', rc, '(cycles since the first wave entered round 0; waves 0, 1 own even rounds, 4, 5 odd ones)')
    for rd in range(8):
        print('round', rd)
        for i, w in enumerate((0, 1, 4, 5)):
            own = (w >> 2) == (rd & 1)
            print('  wave %d %s ' % (w, 'own' if own else '   ') +
                  ' '.join('%s=%d' % (n, v) for n, v in zip(names, st[i, rd, :7]) if own or n != 'G2 done'))
