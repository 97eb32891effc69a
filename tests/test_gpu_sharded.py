"""GPU: two ranks (gloo process group, both on the one visible GPU) shard the particle columns; the
sharded MarkovJumpHMC must reproduce the unsharded device run bit-for-bit -- samples, counters,
evaluation counts -- including when only ONE shard hits a non-finite rate (global retry through
checkpoint + deterministic replay)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, io, contextlib
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from mjhmc_amd.parallel import Comm
from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC, ControlHMC
from mjhmc_amd.misc.distributions import TestGaussian

dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=int(sys.argv[1]), world_size=2)
comm = Comm()
D, N = 24, 301
rs = np.random.RandomState(5)
X0 = rs.randn(D, N)
Xbad = X0.copy()
Xbad[:, 250:] *= 400.0            # only rank 1's columns will overflow exp(H0 - H1): non-finite rate


def dist_of(X):
    class Fixed(TestGaussian):
        def init_X(self):
            self.Xinit = X
    return Fixed(ndims=D, nbatch=N, sigma=1.3)


def run(cls, X, comm, n, **kw):
    d = dist_of(X)
    np.random.seed(9)                                   # the resampling uniforms
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        s = cls(distribution=d, epsilon=kw.pop('eps', 0.3), beta=0.3, num_leapfrog_steps=5, seed=4242, comm=comm, **kw)
        out = s.sample(n)
    return s, d, out, buf.getvalue().count('doubling back')


for name, cls, X, kw in (('mjhmc', MarkovJumpHMC, X0, {}), ('mjhmc-stack', MarkovJumpHMC, X0, dict(resample=False)),
                         ('retry', MarkovJumpHMC, Xbad, dict(eps=1.0)), ('control', ControlHMC, X0, {})):
    s, d, out, nretry = run(cls, X, comm, 7, **dict(kw))
    if comm.rank == 0:
        s1, d1, out1, nretry1 = run(cls, X, None, 7, **dict(kw))
        assert out.shape == out1.shape and np.array_equal(out, out1), name
        assert (s.l_count, s.f_count, s.r_count, s.fl_count) == (s1.l_count, s1.f_count, s1.r_count, s1.fl_count), name
        assert (d.E_count, d.dEdX_count) == (d1.E_count, d1.dEdX_count), name
        assert np.array_equal(s.state.X, s1.state.X) and np.array_equal(s.state.EX, s1.state.EX), name
        assert nretry == nretry1, (name, nretry, nretry1)
        if name == 'retry':
            assert nretry > 0
    else:
        s.state.X, s.state.EX                          # collectives are SPMD: mirror rank 0's gathers
    comm.barrier()

# autocorrelation of a sharded run: every rank transforms its own columns, the lag sums are added
from mjhmc_amd.misc.autocor import calculate_autocorrelation
akw = dict(epsilon=0.3, beta=0.3, num_leapfrog_steps=5, seed=4242, resample=False)
ac, e, g = calculate_autocorrelation(MarkovJumpHMC, dist_of(X0), num_steps=12, comm=comm, **akw)
if comm.rank == 0:
    ac1, e1, g1 = calculate_autocorrelation(MarkovJumpHMC, dist_of(X0), num_steps=12, **akw)
    assert ac.shape == (12,) and np.allclose(ac, ac1, rtol=0, atol=1e-12), 'autocor'
    assert np.array_equal(e, e1) and np.array_equal(g, g1), 'autocor traces'
comm.barrier()
dist.destroy_process_group()
print('rank', sys.argv[1], 'ok')
'''


def test_two_ranks_on_one_gpu_match_unsharded(tmp_path):
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % dict(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (r, out[-4000:])


def test_single_rank_nccl_group(tmp_path):
    """world_size 1 over nccl (RCCL): the collective plumbing the multi-GPU bench uses."""
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%(port)d', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from mjhmc_amd.parallel import Comm
from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
from mjhmc_amd.misc.distributions import TestGaussian
comm = Comm()
assert comm.backend == 'nccl'
np.random.seed(1)
a = MarkovJumpHMC(distribution=TestGaussian(8, 120), epsilon=0.3, beta=0.3, seed=5, comm=comm)
np.random.seed(2); xa = a.sample(5)
np.random.seed(1)
b = MarkovJumpHMC(distribution=TestGaussian(8, 120), epsilon=0.3, beta=0.3, seed=5)
np.random.seed(2); xb = b.sample(5)
assert np.array_equal(xa, xb) and (a.l_count, a.r_count) == (b.l_count, b.r_count)
dist.destroy_process_group()
print('ok')
'''
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    script = tmp_path / 'w1.py'
    script.write_text(code % dict(root=ROOT, port=port))
    p = subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()[-4000:]
