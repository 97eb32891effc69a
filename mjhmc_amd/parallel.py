"""Multi-GPU support: particle columns are independent chains (mjhmc/samplers/hmc_state.py works
column-wise everywhere), so a job shards COLUMNS over ranks -- one process per GPU -- and nothing
is exchanged on the data path.  What crosses ranks (host-side, tiny, or once per ``sample()``):

  * the non-finite-rate flag of an attempt (the reference retries the WHOLE batch,
    markov_jump_hmc.py:376-389) -> min-reduce of "iterations committed", checkpoint + deterministic
    replay on ranks that ran ahead;
  * integer counters (sums);
  * at the end of ``sample()``: the dwell times (to form the global time-major cumsum of
    markov_jump_hmc.py:321-324) and ONE all-gather of the sample columns (RCCL over xGMI when the
    process group's backend is nccl; gloo on CPU for tests).

torch.distributed is plumbing here (process group, collectives); it is imported lazily so the
single-GPU product path never touches torch.
"""
import numpy as np


class ShardPlan(object):
    """Contiguous column blocks: rank r owns [offset[r], offset[r] + count[r])."""

    def __init__(self, n_total, world):
        base, extra = divmod(int(n_total), int(world))
        self.n_total = int(n_total)
        self.world = int(world)
        self.counts = np.array([base + (1 if r < extra else 0) for r in range(world)], dtype=np.int64)
        self.offsets = np.concatenate([[0], np.cumsum(self.counts)[:-1]]).astype(np.int64)

    def span(self, rank):
        return int(self.offsets[rank]), int(self.offsets[rank] + self.counts[rank])

    def owner_of(self, cols):
        """rank owning each global column index."""
        return (np.searchsorted(self.offsets, np.asarray(cols), side='right') - 1).astype(np.int64)


class Comm(object):
    """Thin wrapper over a torch.distributed process group (nccl == RCCL on ROCm, or gloo)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised (launch with torch.distributed.run)')
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device = torch.device('cuda', torch.cuda.current_device()) if self.backend == 'nccl' else torch.device('cpu')

    def _t(self, arr, dtype):
        return self.torch.as_tensor(np.ascontiguousarray(arr), dtype=dtype).to(self.device)

    def allreduce_ints(self, values, op='sum'):
        t = self._t(np.asarray(values, dtype=np.int64), self.torch.int64)
        ops = {'sum': self.dist.ReduceOp.SUM, 'min': self.dist.ReduceOp.MIN, 'max': self.dist.ReduceOp.MAX}
        self.dist.all_reduce(t, op=ops[op], group=self.group)
        return t.cpu().numpy()

    def allreduce_f64(self, values):
        t = self._t(np.asarray(values, dtype=np.float64), self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def bcast(self, arr, src=0):
        t = self._t(arr, self.torch.float64)
        self.dist.broadcast(t, src=src, group=self.group)
        return t.cpu().numpy()

    def allgather_columns(self, block):
        """block (D, m_r) float64 with rank-dependent m_r -> list of the world's blocks.
        ONE collective on a padded tensor (pieces differ in size)."""
        block = np.ascontiguousarray(block, dtype=np.float64)
        D = block.shape[0]
        sizes = self.allreduce_sizes(block.shape[1])
        m_max = int(sizes.max())
        pad = np.zeros((D, m_max))
        pad[:, :block.shape[1]] = block
        mine = self._t(pad, self.torch.float64)
        outs = [self.torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(outs, mine, group=self.group)
        return [o.cpu().numpy()[:, :int(sizes[r])] for r, o in enumerate(outs)]

    def allreduce_sizes(self, m):
        v = np.zeros(self.world, dtype=np.int64)
        v[self.rank] = m
        return self.allreduce_ints(v, 'sum')

    def barrier(self):
        self.dist.barrier(group=self.group)


def agree_on_progress(comm, n_done_local):
    """Iterations every rank has committed == min over ranks (global-batch retry semantics)."""
    return int(comm.allreduce_ints([n_done_local], 'min')[0])


def gather_state_columns(comm, plan, local):
    """(D, N_local) per rank -> (D, N_total) on every rank, global column order."""
    return np.concatenate(comm.allgather_columns(local), axis=1)


def gather_vector(comm, plan, local):
    """(N_local,) per rank -> (N_total,) on every rank."""
    return gather_state_columns(comm, plan, np.asarray(local, dtype=np.float64).reshape(1, -1))[0]


def assemble_stacked(comm, plan, local, n, preserve_order):
    """HMCBase.sample / resample=False (markov_jump_hmc.py:166-173,331-338) from per-rank rings.
    local: (D, N_local, n) if preserve_order else (D, n * N_local) time-major."""
    D = local.shape[0]
    n_loc = int(plan.counts[comm.rank])
    cube = local if preserve_order else local.reshape(D, n, n_loc).transpose(0, 2, 1)      # (D, N_local, n)
    parts = comm.allgather_columns(np.ascontiguousarray(cube).reshape(D, n_loc * n))
    cubes = [p.reshape(D, int(plan.counts[r]), n) for r, p in enumerate(parts)]
    full = np.concatenate(cubes, axis=1)                                                       # (D, N_total, n)
    if preserve_order:
        return full
    return np.ascontiguousarray(full.transpose(0, 2, 1)).reshape(D, n * plan.n_total)


def assemble_resample(comm, plan, n_samples, dwell_local, gather_local, uniforms=None):
    """Dwell-time-weighted resampling of ContinuousTimeHMC.sample (markov_jump_hmc.py:309-329) over
    sharded columns.

    dwell_local : (n_samples, N_local) dwelling times recorded by this rank
    gather_local: f(local pool indices t * N_local + c) -> (D, m) columns from this rank's sample ring
    uniforms    : the n_samples * N_total numbers of ``np.random.random`` (drawn on rank 0 and broadcast
                  when None), so every rank forms the same indices.
    Returns (resamples (D, n_samples * N_total), global pool indices)."""
    N = plan.n_total
    n_loc = int(plan.counts[comm.rank])
    lo = int(plan.offsets[comm.rank])
    # global time-major dwell vector, exactly np.concatenate(dwell_t_k) of the unsharded run
    dwell = gather_state_columns(comm, plan, np.ascontiguousarray(dwell_local))               # (n_samples, N)
    dwell_t = dwell.reshape(-1)
    total_t = np.sum(dwell_t)
    cumul_t = np.cumsum(dwell_t)
    if uniforms is None:
        uniforms = np.random.random(n_samples * N) if comm.rank == 0 else np.zeros(n_samples * N)
        uniforms = comm.bcast(uniforms, src=0)
    rand_vals = np.sort(uniforms) * total_t
    sample_idx = np.searchsorted(cumul_t, rand_vals, side='right')
    if sample_idx.size and sample_idx[-1] >= dwell_t.size:
        raise IndexError('index 0 is out of bounds for axis 0 with size 0')                    # infinite dwell time
    t_of, col_of = np.divmod(sample_idx, N)
    mine = np.nonzero((col_of >= lo) & (col_of < lo + n_loc))[0]
    block = gather_local(t_of[mine] * n_loc + (col_of[mine] - lo))
    parts = comm.allgather_columns(block)                                                       # the one sample gather
    owner = plan.owner_of(col_of)
    out = np.empty((block.shape[0], sample_idx.size))
    for r, part in enumerate(parts):
        out[:, np.nonzero(owner == r)[0]] = part
    return out, sample_idx
