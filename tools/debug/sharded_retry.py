"""Repro of the late-workgroup tally race: two ranks share the GPU, rank 1 fails, E_count of the sharded run
against the unsharded one, four times.  usage: python tools/debug/sharded_retry.py <rank> <port>   (both ranks, same port)"""
import io, contextlib, sys, os
import numpy as np
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mjhmc_amd.parallel import Comm
from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
from mjhmc_amd.misc.distributions import TestGaussian
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%s' % sys.argv[2], rank=int(sys.argv[1]), world_size=2)
comm = Comm()
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
def run(comm):
    d = dist_of(Xbad)
    np.random.seed(9)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        s = MarkovJumpHMC(distribution=d, epsilon=1.0, beta=0.3, num_leapfrog_steps=5, seed=4242, comm=comm)
        out = s.sample(7)
    return s, d, buf.getvalue()
for rep in range(4):
    s, d, log = run(comm)
    if comm.rank == 0:
        print('sharded   E', d.E_count, d.dEdX_count, 'lfr', s.l_count, s.f_count, s.r_count, 'retries', log.count('doubling back'), flush=True)
        s1, d1, log1 = run(None)
        print('unsharded E', d1.E_count, d1.dEdX_count, 'lfr', s1.l_count, s1.f_count, s1.r_count, 'retries', log1.count('doubling back'), flush=True)
    comm.barrier()
dist.destroy_process_group()
