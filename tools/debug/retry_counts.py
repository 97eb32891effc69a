import io, contextlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
from mjhmc_amd.misc.distributions import TestGaussian
D, N = 24, 301
rs = np.random.RandomState(5)
X0 = rs.randn(D, N)
Xbad = X0.copy()
Xbad[:, 250:] *= 400.0
def dist_of(X):
    class Fixed(TestGaussian):
        def init_X(self):
            self.Xinit = X
    return Fixed(ndims=D, nbatch=N, sigma=1.3)
for kw in ({}, dict(preserve_order=True)):
    d = dist_of(Xbad)
    np.random.seed(9)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        s = MarkovJumpHMC(distribution=d, epsilon=1.0, beta=0.3, num_leapfrog_steps=5, seed=4242)
        out = s.sample(7, **kw)
    print(kw, 'retries', buf.getvalue().count('doubling back'), 'counts', s.l_count, s.f_count, s.r_count, 'E', d.E_count, d.dEdX_count, 'eps', s.epsilon)
