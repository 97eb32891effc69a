"""Joins the cycle stamps of the timing builds 30t1 .. 30t6 of the SparseImageCode kernel (one stamp position per build:
a stamp costs ~150 cycles, seven per round would distort what they measure).  Each build also stamps the entry into
round 0; the files are aligned on that.   usage: python tools/sic_stamps_merge.py gpurun_out/stamps_t{1..6}.npy"""
import sys

import numpy as np

names = ['start', 'G2 done', 'at A', 'past A', 'G1 done', 'past B', 'issued']
st = np.zeros((4, 8, 7), dtype=np.int64)
for f in sys.argv[1:]:
    a = np.load(f)
    k = int(f.rsplit('_t', 1)[1].split('.')[0])
    st[:, :, k] = a[:, :, k]
    if k == 1:
        st[:, :, 0] = a[:, :, 0]
for rd in range(8):
    print('round', rd)
    for i, w in enumerate((0, 1, 4, 5)):
        own = (w >> 2) == (rd & 1)
        print('  wave %d %s ' % (w, 'own' if own else '   ') +
              ' '.join('%s=%d' % (names[k], st[i, rd, k]) for k in range(7)
                       if (own or k != 1)))
