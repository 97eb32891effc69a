"""What clock and power the chip holds under each dense / elementwise workload: a thread keeps the sampler iterating for
~16 s while the main thread reads rocm-smi every 2 s.  Evidence behind the "power-throttled" remarks of DESIGN.md 3.5.
usage (GPU box): python tools/clock_sample.py c5 c3 c2 c4"""
import os
import re
import subprocess
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mjhmc_amd import engine, _lib              # noqa: E402
import bench                                    # noqa: E402

ctx = engine.context(0)
for key in sys.argv[1:] or ['c5']:
    w = dict(bench.WORKLOADS[key])
    kind = {'iso': _lib.E_ISO_GAUSS, 'funnel': _lib.E_FUNNEL_NEAL, 'pot': _lib.E_PRODUCT_OF_T, 'sic': _lib.E_SPARSE_CODE}[w['kind']]
    params = w['params']
    if w['kind'] == 'pot':
        W, lognu = bench.pot_model(w['D'])
        params = np.concatenate([[float(w['D'])], W.ravel(), np.exp(lognu), np.zeros(w['D'])])
    if w['kind'] == 'sic':
        B, y, _ = bench.sic_model()
        params = np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), y])
    en = engine.DeviceEnergy(ctx, kind, w['D'], params)
    smp = engine.DeviceSampler(en, bench.initial_state(w, w['N'], 0), seed=1, first_particle_id=0, dtype=w['dtype'])
    smp.set_hparams(w['eps'], w['L'], -np.log(1 - w['beta']) * 0.5, 1.0)
    smp.iterate(4)
    smp.sync()
    stop, count = [False], [0]

    def work():
        while not stop[0]:
            smp.iterate(16)
            smp.sync()
            count[0] += 16

    th = threading.Thread(target=work)
    t0 = time.perf_counter()
    th.start()
    rows = []
    time.sleep(3.0)
    for _ in range(6):
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
        pw = re.search(r'Package Power \(W\): ([\d.]+)', out)
        sclk = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', out)
        rows.append((pw.group(1) if pw else '?', sclk.group(1) if sclk else '?'))
        time.sleep(2.0)
    stop[0] = True
    th.join()
    dt = time.perf_counter() - t0
    print('%s: %.3f ms per iteration over %.0f s; (socket W, sclk MHz) every 2 s: %s' % (key, 1e3 * dt / max(count[0], 1), dt, rows))
    del smp, en
