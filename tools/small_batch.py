"""Per-iteration time of small batches (the reference's typical sizes): fused vs one launch per iteration."""
import os as _os
# the environment A/B switches exist only in the test build of the library (csrc/Makefile: test_hooks)
_os.environ.setdefault('MJHMC_HIP_LIB', _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'mjhmc_amd', 'lib', 'libmjhmc_hip_test.so'))
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib          # noqa: E402

ctx = engine.context(0)
for kind, params, name in ((_lib.E_ISO_GAUSS, [1.0], 'iso'), (_lib.E_FUNNEL_NEAL, [3.0], 'funnel')):
    for D, N in ((2, 100), (2, 1000), (2, 100000), (4, 1000), (10, 1000), (36, 1000), (100, 1000), (32, 10000)):
        X0 = np.random.RandomState(0).randn(D, N)
        smp = engine.DeviceSampler(engine.DeviceEnergy(ctx, kind, D, params), X0, seed=1)
        smp.set_hparams(0.05, 10, 0.05, 1.0, 0.5)
        out = []
        for env in (None, '1'):
            if env:
                os.environ['MJHMC_NO_FUSE'] = env
            else:
                os.environ.pop('MJHMC_NO_FUSE', None)
            smp.iterate(256)
            t0 = time.perf_counter()
            smp.iterate(1024)
            smp.sync()
            out.append((time.perf_counter() - t0) / 1024 * 1e6)
            out.append(smp.last_timing()['total_ms'] / 1024 * 1e3)
        print('%-6s D=%3d N=%5d  default %.1f us/iter (device %.1f)   one launch per iteration %.1f us/iter (device %.1f)' % (name, D, N, out[0], out[1], out[2], out[3]), flush=True)
        smp.close()

# single-iteration CALLS (what a loop of sampling_iteration() makes): call overhead included
print('-- one mjhmc_iterate call per iteration (wall clock per call) --')
for kind, params, name in ((_lib.E_ISO_GAUSS, [1.0], 'iso'), (_lib.E_FUNNEL_NEAL, [3.0], 'funnel')):
    for D, N in ((2, 100), (2, 1000), (10, 1000), (32, 10000)):
        X0 = np.random.RandomState(0).randn(D, N)
        smp = engine.DeviceSampler(engine.DeviceEnergy(ctx, kind, D, params), X0, seed=1)
        smp.set_hparams(0.05, 10, 0.05, 1.0, 0.5)
        os.environ.pop('MJHMC_NO_FUSE', None)
        for _ in range(200):
            smp.iterate(1)
        t0 = time.perf_counter()
        for _ in range(2000):
            smp.iterate(1)
        smp.sync()
        print('%-6s D=%3d N=%5d  %.1f us per call' % (name, D, N, (time.perf_counter() - t0) / 2000 * 1e6), flush=True)
        smp.close()
