"""CPU, world_size 2 over gloo: the N > 1 host path (mjhmc_amd/parallel.py) -- shard plan, the
global-retry agreement, and the dwell-time resampling / stacking assembled from per-rank shards must
reproduce the unsharded result bit-for-bit.  Each rank stands in for its GPU with the NumPy oracle run
on its own columns (the RNG is keyed by global particle id, like the device's)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from mjhmc_amd.parallel import Comm, ShardPlan, agree_on_progress, assemble_resample, assemble_stacked, gather_state_columns, gather_vector
from oracle import mjhmc_oracle as orc

dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=int(sys.argv[1]), world_size=2)
comm = Comm()
D, N, n, eps, L, beta, seed = 5, 37, 6, 0.4, 4, 0.4, 777
plan = ShardPlan(N, comm.world)
assert plan.counts.tolist() == [19, 18] and plan.offsets.tolist() == [0, 19]
assert plan.owner_of([0, 18, 19, 36]).tolist() == [0, 0, 1, 1]
lo, hi = plan.span(comm.rank)
X0 = np.random.RandomState(3).randn(D, N)
u = np.random.RandomState(4).rand(n * N)


class FixedU(orc.PhiloxRNG):
    def uniforms(self, k):
        return u[:k].copy()


def run(cols):
    s = orc.MarkovJumpHMC(orc.IsoGaussian(1.0), X0[:, cols], epsilon=eps, beta=beta, num_leapfrog_steps=L,
                          rng=FixedU(seed, cols))
    kept, dwell = [], []
    s.sampling_iteration(); kept.append(s.state.X.copy())
    for _ in range(n):
        dwell.append(s.dwelling_times.copy()); s.sampling_iteration(); kept.append(s.state.X.copy())
    return s, kept, np.stack(dwell)


# --- unsharded truth (both ranks compute it) ---------------------------------------------------
ref = orc.MarkovJumpHMC(orc.IsoGaussian(1.0), X0, epsilon=eps, beta=beta, num_leapfrog_steps=L, rng=FixedU(seed, np.arange(N)))
want = ref.sample(n)

# --- sharded -------------------------------------------------------------------------------------
s, kept, dwell = run(np.arange(lo, hi))
pool = np.concatenate(kept[:-1], axis=1)                       # local time-major pool (D, n * N_local)
got, idx = assemble_resample(comm, plan, n, dwell, lambda i: pool[:, i], uniforms=u)
assert got.shape == want.shape and np.array_equal(got, want), 'resampled columns differ'
assert np.array_equal(idx, ref.last_pick)

# stacked outputs (resample=False): the last n states
ref2 = orc.MarkovJumpHMC(orc.IsoGaussian(1.0), X0, epsilon=eps, beta=beta, num_leapfrog_steps=L, resample=False,
                         rng=orc.PhiloxRNG(seed, np.arange(N)))
want_cat, = [ref2.sample(n)]
s2 = orc.MarkovJumpHMC(orc.IsoGaussian(1.0), X0[:, lo:hi], epsilon=eps, beta=beta, num_leapfrog_steps=L, resample=False,
                       rng=orc.PhiloxRNG(seed, np.arange(lo, hi)))
loc_cat = s2.sample(n)
assert np.array_equal(assemble_stacked(comm, plan, loc_cat, n, False), want_cat)
loc_cube = loc_cat.reshape(D, n, hi - lo).transpose(0, 2, 1)
want_cube = want_cat.reshape(D, n, N).transpose(0, 2, 1)
assert np.array_equal(assemble_stacked(comm, plan, np.ascontiguousarray(loc_cube), n, True), want_cube)

# state / vector gathers and the retry agreement
assert np.array_equal(gather_state_columns(comm, plan, s2.state.X), ref2.state.X)
assert np.array_equal(gather_vector(comm, plan, s2.dwelling_times), ref2.dwelling_times)
assert agree_on_progress(comm, 5 if comm.rank == 0 else 3) == 3
assert comm.allreduce_ints([1, comm.rank], 'sum').tolist() == [2, 1]
assert np.array_equal(comm.bcast(np.arange(4.0) if comm.rank == 0 else np.zeros(4)), np.arange(4.0))
dist.barrier()
dist.destroy_process_group()
print('rank', sys.argv[1], 'ok')
'''


def test_two_rank_assembly_matches_unsharded(tmp_path):
    import socket
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
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (r, out[-3000:])
        assert 'ok' in out


def test_shard_plan_edges():
    from mjhmc_amd.parallel import ShardPlan
    p = ShardPlan(10, 4)
    assert p.counts.tolist() == [3, 3, 2, 2] and p.offsets.tolist() == [0, 3, 6, 8]
    assert p.owner_of(np.arange(10)).tolist() == [0, 0, 0, 1, 1, 1, 2, 2, 3, 3]
    assert ShardPlan(8, 8).counts.tolist() == [1] * 8
    assert ShardPlan(100000, 8).span(7) == (87500, 100000)
