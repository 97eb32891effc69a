"""Time the ring-resident autocorrelation against the NumPy restatement (same samples).
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
    if ring_gb <= 4:
        from oracle import autocor_oracle as aco
        t0 = time.time()
        samples = smp._stack(T, True)
        t_dl = time.time() - t0
        t0 = time.time()
        ref = aco.fft_autocor(samples)
        t_np = time.time() - t0
        print('download %.2f s  numpy fft_autocor %.2f s  max|diff| %.2e' % (t_dl, t_np, np.abs(sums / sums[0] - ref).max()))


if __name__ == '__main__':
    main()
