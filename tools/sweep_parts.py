"""Dense workloads: free-running parts of a batch (api.hip: iterate_t) -- 1 ... 4 parts at full size and at N/8.
usage: python tools/sweep_parts.py [c3f64|c3|c5|c5bf16 ...]      (test build: MJHMC_SPLIT_PARTS / MJHMC_NO_SPLIT)"""
import os as _os
_os.environ.setdefault('MJHMC_HIP_LIB', _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'mjhmc_amd', 'lib', 'libmjhmc_hip_test.so'))
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                   # noqa: E402
from mjhmc_amd import engine, _lib             # noqa: E402


def main():
    keys = sys.argv[1:] or ['c3f64', 'c5']
    ctx = engine.context(0)
    for key in keys:
        w = bench.WORKLOADS[key]
        for shard in (1, 8):
            n = w['N'] // shard
            if w['kind'] == 'pot':
                W, lognu = bench.pot_model(w['D'])
                params = np.concatenate([[float(w['D'])], W.ravel(), np.exp(lognu), np.zeros(w['D'])])
                kind = _lib.E_PRODUCT_OF_T
            else:
                B, y, _ = bench.sic_model()
                params = np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), y])
                kind = _lib.E_SPARSE_CODE
            en = engine.DeviceEnergy(ctx, kind, w['D'], params)
            smp = engine.DeviceSampler(en, bench.initial_state(w, n, 0), seed=1, dtype=w['dtype'])
            smp.set_hparams(w['eps'], w['L'], -np.log(1 - w['beta']) * 0.5, 1.0)
            smp.iterate(4)
            row = []
            for parts in ('1', '2', '3', '4'):
                if parts == '1':
                    os.environ['MJHMC_NO_SPLIT'] = '1'
                else:
                    os.environ.pop('MJHMC_NO_SPLIT', None)
                    os.environ['MJHMC_SPLIT_PARTS'] = parts
                smp.iterate(8)
                smp.sync()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    smp.iterate(16)
                    smp.sync()
                    best = min(best, (time.perf_counter() - t0) * 1e3 / 16)
                row.append('%s parts %.3f ms' % (parts, best))
            os.environ.pop('MJHMC_NO_SPLIT', None)
            os.environ.pop('MJHMC_SPLIT_PARTS', None)
            print('%-6s N=%7d: %s' % (key, n, ' | '.join(row)), flush=True)
            smp.close()


if __name__ == '__main__':
    main()
