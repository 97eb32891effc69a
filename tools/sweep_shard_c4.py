"""C4 (Neal funnel, ndims 32, L 15, float64) at shard sizes: the compacted three-launch iteration against fused launches.
usage: python tools/sweep_shard_c4.py [steps]      (test build: MJHMC_FUSE_BELOW moves the fused / compacted threshold, MJHMC_NO_ROWS)"""
import os as _os
_os.environ.setdefault('MJHMC_HIP_LIB', _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'mjhmc_amd', 'lib', 'libmjhmc_hip_test.so'))
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib          # noqa: E402


def time_ms(smp, n, reps=8):
    smp.iterate(n)
    smp.sync()
    ev, wall = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        smp.iterate(n)
        smp.sync()
        wall.append((time.perf_counter() - t0) * 1e3 / n)
        ev.append(smp.last_timing()['total_ms'] / n)
    return float(np.median(ev)), float(np.median(wall))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ctx = engine.context(0)
    full = None
    for N in (1000000, 500000, 250000, 125000, 62500):
        rng = np.random.RandomState(0)
        X0 = rng.randn(32, N)
        X0[0] *= 3.0
        X0[1:] *= np.exp(X0[0] / 2.)
        en = engine.DeviceEnergy(ctx, _lib.E_FUNNEL_NEAL, 32, [3.0])
        smp = engine.DeviceSampler(en, X0, seed=1)
        smp.set_hparams(0.05, 15, -np.log(0.9) * 0.5, 1.0)
        row = {}
        # compacted: trajectory launch (row form) + jump-process launch; groups: the same with a group of lanes per particle;
        # fused: all iterations of a call in one launch, row form; fused_groups: the same with a group of lanes per particle
        for tag, env in (('compacted', '0'), ('groups', '0'), ('fused', '100000000'), ('fused_groups', '100000000')):
            os.environ['MJHMC_FUSE_BELOW'] = env
            if tag.endswith('groups'):
                os.environ['MJHMC_NO_ROWS'] = '1'
            else:
                os.environ.pop('MJHMC_NO_ROWS', None)
            row[tag] = time_ms(smp, steps)
        if full is None:
            full = min(r[1] for r in row.values())
        g = 1000000 // N
        print('N=%8d steps=%d  ' % (N, steps) + '  '.join('%s %.4f (wall %.4f)' % (t, row[t][0], row[t][1]) for t in row)
              + '   shard efficiency at G=%d: ' % g + ' '.join('%s %.2f' % (t, full / (g * row[t][1])) for t in row), flush=True)
        smp.close()


if __name__ == '__main__':
    main()
