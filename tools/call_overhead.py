"""What a sampling_iteration()-style caller waits per call at C4's size: wall clock against the launches' HIP-event time,
and where the host side of it goes (cProfile of the Python layer)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib  # noqa: E402

ctx = engine.context(0)
en = engine.DeviceEnergy(ctx, _lib.E_FUNNEL_NEAL, 32, [3.0])
X0 = np.random.RandomState(0).randn(32, 1000000)
s = engine.DeviceSampler(en, X0, seed=1)
s.set_hparams(0.05, 15, 0.05, 1.0)
s.iterate(3)
for _ in range(20):
    s.iterate(1)
s.sync()
t = time.perf_counter()
for _ in range(200):
    s.iterate(1)
wall = (time.perf_counter() - t) / 200 * 1e3
try:
    print('wall per call ms', wall, 'launches (events)', s.last_timing())
except Exception:          # (nothing recorded)
    print('wall per call ms', wall)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    s.iterate(1)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
