import os as _os
# the environment A/B switches exist only in the test build of the library (csrc/Makefile: test_hooks)
_os.environ.setdefault('MJHMC_HIP_LIB', _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'mjhmc_amd', 'lib', 'libmjhmc_hip_test.so'))
import os, sys
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from mjhmc_amd import engine, _lib
ctx = engine.context(0)
for kind, params, name in ((_lib.E_ISO_GAUSS, [1.0], 'iso'), (_lib.E_FUNNEL_NEAL, [3.0], 'funnel')):
  for D, N in ((16, 3200000), (32, 1600000), (64, 800000), (128, 400000)):
    X0 = np.random.RandomState(0).randn(D, N)
    if name == 'funnel':
        X0[0] *= 3; X0[1:] *= np.exp(X0[0] / 2.)
    for C in ('4', '1'):
        os.environ['MJHMC_CHUNKS_PER_LANE'] = C
        smp = engine.DeviceSampler(engine.DeviceEnergy(ctx, kind, D, params), X0, seed=1)
        smp.set_hparams(0.05, 10, 0.0527, 1.0, 0.5)
        smp.iterate(64)
        best = 1e9
        for _ in range(3):
            st, done = smp.iterate(64)
            best = min(best, smp.last_timing()['total_ms'] / 64)
        print('%-6s D=%3d N=%7d chunks/lane %s: %.4f ms/iter (done %d)' % (name, D, N, C, best, done), flush=True)
        smp.close()
