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

# ProductOfT keeps dE/dX as part of the state: a rank that ran ahead of a failure elsewhere must get it back too.
# Huge initial momenta on rank 1's columns only: the leapfrog energy error there exceeds log(DBL_MAX) at the
# nominal step size (a non-finite rate) and shrinks as the retry halves it.
from mjhmc_amd.misc.distributions import ProductOfT
rs2 = np.random.RandomState(8)
Dp, Np = 36, 120
sp = rs2.rand(Dp, Dp); Wp = rs2.randn(Dp, Dp); Wp[sp > 0.05] = 0; Wp += np.eye(Dp)
lognu = np.log(rs2.rand(Dp) * 2 + 2.1)
Xp = rs2.randn(Dp, Np)
Vp = rs2.randn(Dp, Np)
Vp[:, 100:] *= 3000.0


def run_pot(comm):
    class FixedT(ProductOfT):
        def init_X(self):
            self.Xinit = Xp
    d = FixedT(ndims=Dp, nbasis=Dp, nbatch=Np, lognu=lognu, W=Wp, state_dtype='float32')
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        s = MarkovJumpHMC(distribution=d, epsilon=0.1, beta=0.3, num_leapfrog_steps=6, seed=99, comm=comm,
                          resample=False, Vinit=Vp)
        out = s.sample(6)
    return s, d, out, buf.getvalue().count('doubling back')


s, d, out, nretry = run_pot(comm)
if comm.rank == 0:
    s1, d1, out1, nretry1 = run_pot(None)
    assert nretry1 > 0, 'the ProductOfT case no longer provokes a retry'
    assert nretry == nretry1 and np.array_equal(out, out1), ('pot', nretry, nretry1)
    assert (s.l_count, s.f_count, s.r_count) == (s1.l_count, s1.f_count, s1.r_count)
    assert (d.E_count, d.dEdX_count) == (d1.E_count, d1.dEdX_count)
    assert np.array_equal(s.state.X, s1.state.X) and np.array_equal(s.state.V, s1.state.V)
else:
    s.state.X, s.state.V
comm.barrier()

# mjhmc_rollback (single iterations: sampling_iteration() calls) on the samplers that do NOT ping-pong their state -- the
# multi-pass path (rows wider than the register kernels hold) commits in place and leaves the pre-move state in its
# proposal workspace -- and on the float64-state ProductOfT tile kernel (ping-pong parities, stored dE/dX included).
# The failure sits on rank 1's columns only, so rank 0 has committed the iteration every time the batch retries.
Dw, Nw = 1030, 90
Xw = np.random.RandomState(12).randn(Dw, Nw)
Xw[:, 70:] *= 60.0


def steps(make, comm, n):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        s, d = make(comm)
        for _ in range(n):
            s.sampling_iteration()
    return s, d, buf.getvalue().count('doubling back')


def make_wide(comm):
    class FixedW(TestGaussian):
        def init_X(self):
            self.Xinit = Xw
    d = FixedW(ndims=Dw, nbatch=Nw, sigma=1.3)
    return MarkovJumpHMC(distribution=d, epsilon=0.5, beta=0.3, num_leapfrog_steps=4, seed=77, comm=comm, resample=False), d


def make_pot64(comm):
    class FixedT(ProductOfT):
        def init_X(self):
            self.Xinit = Xp
    d = FixedT(ndims=Dp, nbasis=Dp, nbatch=Np, lognu=lognu, W=Wp, state_dtype='float64')
    return MarkovJumpHMC(distribution=d, epsilon=0.1, beta=0.3, num_leapfrog_steps=6, seed=99, comm=comm, resample=False,
                         Vinit=Vp), d


for name, make in (('wide', make_wide), ('pot64', make_pot64)):
    s, d, nretry = steps(make, comm, 4)
    if comm.rank == 0:
        s1, d1, nretry1 = steps(make, None, 4)
        assert nretry1 > 0, name + ': the case no longer provokes a retry'
        assert nretry == nretry1, (name, nretry, nretry1)
        assert (s.l_count, s.f_count, s.r_count) == (s1.l_count, s1.f_count, s1.r_count), name
        assert (d.E_count, d.dEdX_count) == (d1.E_count, d1.dEdX_count), name
        assert np.array_equal(s.state.X, s1.state.X) and np.array_equal(s.state.V, s1.state.V), name
        assert np.array_equal(s.state.EX, s1.state.EX) and np.array_equal(s.state.dEdX, s1.state.dEdX), name
    else:
        s.state.X, s.state.V, s.state.EX, s.state.dEdX
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


def test_rccl_communicator_entry_points(tmp_path):
    """The library's own RCCL communicator (mjhmc_comm_*, include/mjhmc_hip.h) with one rank -- RCCL refuses two ranks
    on one device, and the box has one -- through every entry point: the host-value collectives, and the device-ring
    all-gathers behind sample() on all three output paths, which must reproduce the communicator-free run bit for bit."""
    import numpy as np
    from mjhmc_amd.parallel import RcclComm, ShardPlan
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC, ControlHMC
    from mjhmc_amd.misc.distributions import TestGaussian, SparseImageCode
    comm = RcclComm(0, 1, device=0, id_path=str(tmp_path / 'comm.id'))
    assert comm.world == 1 and comm.on_device and comm.backend == 'rccl'
    assert comm.allreduce_ints([3, -4], 'sum').tolist() == [3, -4] and comm.allreduce_ints([7], 'min').tolist() == [7]
    assert comm.allreduce_f64(np.array([1.5, 2.5]), 'max').tolist() == [1.5, 2.5]
    assert np.array_equal(comm.bcast(np.arange(5.0)), np.arange(5.0))
    blk = np.random.RandomState(0).randn(3, 7)
    parts = comm.allgather_columns(blk)
    assert len(parts) == 1 and np.array_equal(parts[0], blk)
    comm.barrier()

    for cls, kw, po in ((MarkovJumpHMC, {}, False), (MarkovJumpHMC, dict(resample=False), False),
                        (MarkovJumpHMC, dict(resample=False), True), (ControlHMC, {}, True)):
        outs = []
        for c in (comm, None):
            np.random.seed(1)
            d = TestGaussian(ndims=24, nbatch=301, sigma=1.3)
            s = cls(distribution=d, epsilon=0.3, beta=0.3, num_leapfrog_steps=5, seed=4242, comm=c, **kw)
            np.random.seed(2)
            outs.append((s.sample(7, preserve_order=po), s.l_count, s.f_count, s.r_count, d.E_count, d.dEdX_count,
                         s.state.X))
        a, b = outs
        assert a[0].shape == b[0].shape and np.array_equal(a[0], b[0]), (cls.__name__, kw, po)
        assert a[1:6] == b[1:6] and np.array_equal(a[6], b[6])
    # a bf16 ring goes through the same gather (rows are 2 KB of bfloat16)
    from tests.helpers import sic_problem, to_bf16
    B, imgs, a0 = sic_problem(0)
    X0 = to_bf16(a0[:, None] + 0.1 * np.random.RandomState(3).randn(1024, 40))
    outs = []
    for c in (comm, None):
        d = SparseImageCode(n_patches=1, n_batches=40, n_basis=1024, basis=B, imgs=imgs, init=X0, state_dtype='bfloat16')
        s = MarkovJumpHMC(distribution=d, epsilon=0.05, beta=0.2, num_leapfrog_steps=4, seed=6, comm=c, resample=False)
        outs.append(s.sample(3))
    assert np.array_equal(outs[0], outs[1])
    # the root-only form of the data-path collective (mjhmc_comm_allgather_ring with host_out = NULL): a rank that is not
    # the root runs the pack and the collective and keeps nothing; the root's block is the all-ranks form's
    dev = s._dev
    counts = np.array([40], dtype=np.int64)
    full = comm.allgather_ring(dev, 0, 3, False, counts)
    assert np.array_equal(comm.allgather_ring(dev, 0, 3, False, counts, root=0), full)
    assert comm.allgather_ring(dev, 0, 3, False, counts, root=1) is None          # (this rank is 0: not the root)
    # ... and through the sampler: gather_root = this rank -> sample() returns the block, another rank -> None
    d2 = TestGaussian(ndims=24, nbatch=301, sigma=1.3)
    s2 = MarkovJumpHMC(distribution=d2, epsilon=0.3, beta=0.3, num_leapfrog_steps=5, seed=4242, comm=comm, resample=False)
    s2.gather_root = 0
    blk = s2.sample(4, preserve_order=True)
    assert blk.shape == (24, 301, 4)
    s2.gather_root = 1
    assert s2.sample(4, preserve_order=True) is None
    assert np.array_equal(comm.allgather_ring(dev, 0, 3, True, counts, root=0), comm.allgather_ring(dev, 0, 3, True, counts))
    comm.close()


def test_single_rank_nccl_group(tmp_path):
    """world_size 1 over nccl (RCCL) through the torch.distributed shim (kept for the gloo tests)."""
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


@pytest.mark.parametrize('case', ['f64', 'f64_ragged', 'f32_pot', 'bf16_sic'])
def test_gather_unpack_for_ranks_above_zero_on_one_gpu(case):
    """The pack and unpack steps of the two sample all-gathers (csrc/comm.hip: ring_pack / ring_unpack, columns_pack /
    columns_unpack) with the collective replaced by "all ranks' samplers live on this GPU" (test build of the library,
    mjhmc_test_gather_*_local): three UNEVEN column shards (34 / 33 / 33), three ring slots -- the offsets of ranks
    1 and 2, the padding to the largest shard, the time-major and the stacked layout and the rank-major column gather
    with its scatter back into sample order, for float64, float32 and bfloat16 row pitches."""
    import ctypes
    import numpy as np
    from mjhmc_amd import engine, _lib
    from mjhmc_amd.parallel import ShardPlan
    from tests.helpers import hooks_context, sic_problem, ref_init_weights
    ctx = hooks_context(0)
    lib = ctx.lib
    N, n, world = 100, 3, 3
    plan = ShardPlan(N, world)
    assert plan.counts.tolist() == [34, 33, 33]
    rs = np.random.RandomState(12)
    if case in ('f64', 'f64_ragged'):
        D = 24 if case == 'f64' else 5
        en, X0, dtype, hp = engine.DeviceEnergy(ctx, _lib.E_ISO_GAUSS, D, [1.0]), rs.randn(D, N), 'float64', (0.2, 4, 0.1)
    elif case == 'f32_pot':
        D = 36
        W, lognu = ref_init_weights(D, D)
        en = engine.DeviceEnergy(ctx, _lib.E_PRODUCT_OF_T, D, np.concatenate([[float(D)], W.ravel(), np.exp(lognu), np.zeros(D)]))
        X0, dtype, hp = rs.randn(D, N), 'float32', (0.1, 4, 0.1)
    else:
        D = 1024
        B, imgs, a0 = sic_problem(0)
        en = engine.DeviceEnergy(ctx, _lib.E_SPARSE_CODE, D, np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), imgs[:, :1].T.ravel()]))
        X0, dtype, hp = a0[:, None] + 0.1 * rs.randn(D, N), 'bfloat16', (0.0625, 3, 0.1)

    def run(lo, hi):
        s = engine.DeviceSampler(en, X0[:, lo:hi], seed=21, first_particle_id=lo, dtype=dtype)
        s.set_hparams(hp[0], hp[1], hp[2], 1.0)
        s.ring_alloc(n)
        s.iterate(n, ring_slot0=0)
        return s

    shards = [run(*plan.span(r)) for r in range(world)]
    handles = (ctypes.c_void_p * world)(*[s.handle for s in shards])
    per_cat = [s.ring_read(0, n, stacked=False).reshape(D, n, -1) for s in shards]           # (D, n, N_r) per rank
    want_cat = np.concatenate(per_cat, axis=2).reshape(D, n * N)
    want_cube = np.concatenate([s.ring_read(0, n, stacked=True) for s in shards], axis=1)     # (D, N, n)
    got = np.full((D, n * N), np.nan)
    _lib.check(lib.mjhmc_test_gather_ring_local(handles, world, 0, n, 0, _lib.ptr(got)), lib)
    assert np.array_equal(got, want_cat), 'time-major'
    got = np.full((D, N, n), np.nan)
    _lib.check(lib.mjhmc_test_gather_ring_local(handles, world, 0, n, 1, _lib.ptr(got)), lib)
    assert np.array_equal(got, want_cube), 'stacked'
    # a sub-range of slots
    got = np.full((D, 2 * N), np.nan)
    _lib.check(lib.mjhmc_test_gather_ring_local(handles, world, 1, 2, 0, _lib.ptr(got)), lib)
    assert np.array_equal(got, want_cat.reshape(D, n, N)[:, 1:].reshape(D, 2 * N)), 'slots 1..2'

    # resampled columns: global pool indices t * N + c -> owner, local pool index t * N_r + (c - lo), rank-major blocks,
    # scattered back into sample order exactly as parallel.assemble_resample does
    m = 257
    sample_idx = np.sort(rs.randint(0, n * N, size=m))
    t_of, col_of = np.divmod(sample_idx, N)
    owner = plan.owner_of(col_of)
    lists, counts = [], []
    for r in range(world):
        mine = np.nonzero(owner == r)[0]
        lists.append(t_of[mine] * int(plan.counts[r]) + (col_of[mine] - int(plan.offsets[r])))
        counts.append(mine.size)
    idx = np.ascontiguousarray(np.concatenate(lists), dtype=np.int64)
    cnt = np.ascontiguousarray(counts, dtype=np.int64)
    flat = np.full((D, m), np.nan)
    _lib.check(lib.mjhmc_test_gather_columns_local(handles, world, _lib.ptr(idx), _lib.ptr(cnt), _lib.ptr(flat)), lib)
    out = np.empty_like(flat)
    out[:, np.argsort(owner, kind='stable')] = flat
    assert np.array_equal(out, want_cat[:, sample_idx]), 'resampled columns'
    # a rank that owns none of the picked columns
    only0 = np.ascontiguousarray(lists[0][:5], dtype=np.int64)
    flat = np.full((D, 5), np.nan)
    _lib.check(lib.mjhmc_test_gather_columns_local(handles, world, _lib.ptr(only0), _lib.ptr(np.array([5, 0, 0], dtype=np.int64)),
                                                   _lib.ptr(flat)), lib)
    assert np.array_equal(flat, want_cat[:, sample_idx[owner == 0][:5]])

    if dtype == 'float64':      # and the shards ARE the unsharded run's columns (counter RNG keyed by global particle id)
        whole = run(0, N)
        assert np.array_equal(want_cat, whole.ring_read(0, n, stacked=False))
        whole.close()
    for s in shards:
        s.close()
