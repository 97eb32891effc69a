"""Cycle stamps of the relay kernel's sampling iterations (tools/rows_stamps.sh builds the library):
   MJHMC_HIP_LIB=$PWD/mjhmc_amd/lib/librows_stamps.so python tools/rows_stamps.py [N] [L] [iterations]
Prints, per wave of workgroup 0 (first tile), the mean cycles between consecutive stamp points over the iterations of one
call (the first two iterations dropped), and the iteration length."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib          # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 15
n_it = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ctx = engine.context(0)
rng = np.random.RandomState(0)
X0 = rng.randn(32, N)
X0[0] *= 3.0
X0[1:] *= np.exp(X0[0] / 2.)
en = engine.DeviceEnergy(ctx, _lib.E_FUNNEL_NEAL, 32, [3.0])
smp = engine.DeviceSampler(en, X0, seed=1)
smp.set_hparams(0.05, L, -np.log(0.9) * 0.5, 1.0)
for _ in range(3):
    smp.iterate(n_it)
    smp.sync()
print('total_ms per iteration', smp.last_timing()['total_ms'] / n_it)
lib = ctx.lib
buf = np.zeros((4, 64, 16), dtype=np.uint64)
lib.mjhmc_rows_stamps.argtypes = [ctypes.c_void_p]
assert lib.mjhmc_rows_stamps(buf.ctypes.data_as(ctypes.c_void_p)) == 0
names = ['top', 'copies', 'relay: enter', 'relay: waited', 'relay: done', 'own L done', 'energies', 'pool waited', 'decide',
         'successor', 'refresh', 'tally/ring', 'deposit']
st = buf[:, 2:n_it, :13].astype(np.int64)
t0 = st[:, :, 0:1]
rel = st - t0
print('point: mean cycles since the iteration top, per wave (relay stamps belong to the wave\'s own part)')
for k, nm in enumerate(names):
    print('%-16s' % nm, ' '.join('%8.0f' % rel[w, :, k].mean() for w in range(4)))
it_len = np.diff(buf[:, 2:n_it, 0].astype(np.int64), axis=1)
print('%-16s' % 'iteration', ' '.join('%8.0f' % it_len[w].mean() for w in range(4)))
print('relay wait       ', ' '.join('%8.0f' % (st[w, :, 3] - st[w, :, 2]).mean() for w in range(4)))
print('relay work       ', ' '.join('%8.0f' % (st[w, :, 4] - st[w, :, 3]).mean() for w in range(4)))
print('pool wait        ', ' '.join('%8.0f' % (st[w, :, 7] - st[w, :, 6]).mean() for w in range(4)))
print('decide           ', ' '.join('%8.0f' % (st[w, :, 8] - st[w, :, 7]).mean() for w in range(4)))
print('successor        ', ' '.join('%8.0f' % (st[w, :, 9] - st[w, :, 8]).mean() for w in range(4)))
print('refresh          ', ' '.join('%8.0f' % (st[w, :, 10] - st[w, :, 9]).mean() for w in range(4)))
print('tally            ', ' '.join('%8.0f' % (st[w, :, 11] - st[w, :, 10]).mean() for w in range(4)))
print('deposit          ', ' '.join('%8.0f' % (st[w, :, 12] - st[w, :, 11]).mean() for w in range(4)))

# the tile's fixed part (first workgroup tile): rows + scalars in (13 -> 14), first iteration top (14 -> it 0's stamp 0),
# last iteration's end -> rows out and stored (stamp 12 of the last iteration -> 15)
b = buf.astype(np.int64)
print('tile: load      ', ' '.join('%8.0f' % (b[w, 0, 14] - b[w, 0, 13]) for w in range(4)))
print('tile: to it 0   ', ' '.join('%8.0f' % (b[w, 0, 0] - b[w, 0, 14]) for w in range(4)))
print('tile: store     ', ' '.join('%8.0f' % (b[w, 0, 15] - b[w, n_it - 1, 12]) for w in range(4)))
print('tile: whole     ', ' '.join('%8.0f' % (b[w, 0, 15] - b[w, 0, 13]) for w in range(4)), ' = %d iterations + the fixed part' % n_it)
