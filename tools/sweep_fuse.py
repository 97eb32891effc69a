"""Fused vs one-iteration-per-launch jump kernels over state-row sizes (A/B inside one process).
usage: python tools/sweep_fuse.py"""
import os as _os
# the environment A/B switches exist only in the test build of the library (csrc/Makefile: test_hooks)
_os.environ.setdefault('MJHMC_HIP_LIB', _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'mjhmc_amd', 'lib', 'libmjhmc_hip_test.so'))
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib          # noqa: E402


def time_ms(smp, n):
    smp.iterate(n)
    best = 1e9
    for _ in range(3):
        smp.iterate(n)
        best = min(best, smp.last_timing()['total_ms'] / n)
    return best


def main():
    ctx = engine.context(0)
    total = 512 * 100000
    for kind, params, name in ((_lib.E_ISO_GAUSS, [1.0], 'iso'), (_lib.E_FUNNEL_NEAL, [3.0], 'funnel')):
        for D in (8, 16, 32, 64, 128, 256, 512, 1024):
            N = total // D
            X0 = np.random.RandomState(0).randn(D, N)
            en = engine.DeviceEnergy(ctx, kind, D, params)
            smp = engine.DeviceSampler(en, X0, seed=1)
            for L in (5, 10, 20):
                smp.set_hparams(0.05, L, 0.05, 1.0, 0.5)
                os.environ.pop('MJHMC_NO_FUSE', None)
                f = time_ms(smp, 40)
                os.environ['MJHMC_NO_FUSE'] = '1'
                u = time_ms(smp, 40)
                print('%-6s D=%4d N=%8d L=%2d  fused %.4f ms  unfused %.4f ms  ratio %.2f' % (name, D, N, L, f, u, u / f), flush=True)
            smp.close()


if __name__ == '__main__':
    main()
