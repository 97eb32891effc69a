"""Time the ring-resident autocorrelation (the comparison with the NumPy restatement lives in
tests/test_gpu_autocor.py; the oracle is test infrastructure and is not imported from tools/).
usage: python tools/autocor_bench.py [D N T]"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC          # noqa: E402
from mjhmc_amd.misc.distributions import TestGaussian                  # noqa: E402


def main():
    D, N, T = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (100, 1000, 2000)
    X0 = np.random.RandomState(0).randn(D, N)

    class Fixed(TestGaussian):
        def init_X(self):
            self.Xinit = X0
    smp = MarkovJumpHMC(distribution=Fixed(ndims=D, nbatch=N, sigma=1.0), epsilon=0.3, beta=0.1, num_leapfrog_steps=5,
                        seed=1, resample=False)
    t0 = time.time()
    smp._record(T)
    smp._dev.sync()
    t_run = time.time() - t0
    times = []
    for _ in range(3):
        t0 = time.time()
        sums = smp._dev.ring_autocor(0, T)
        times.append(time.time() - t0)
    ring_gb = D * N * T * 8 / 1e9
    print('ring %.2f GB  sampling %.3f s  autocor calls %s s  -> %.1f GB/s of ring' %
          (ring_gb, t_run, ['%.3f' % t for t in times], ring_gb / min(times)))
    assert abs(sums[0]) > 0 and np.isfinite(sums).all()


if __name__ == '__main__':
    main()
