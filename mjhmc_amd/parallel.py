"""Multi-GPU support: particle columns are independent chains (mjhmc/samplers/hmc_state.py works
column-wise everywhere), so a job shards COLUMNS over ranks -- one process per GPU -- and nothing
is exchanged on the data path.  What crosses ranks (tiny, or once per ``sample()``):

  * the non-finite-rate flag of an attempt (the reference retries the WHOLE batch,
    markov_jump_hmc.py:376-389) -> min-reduce of "iterations committed", rollback + deterministic
    replay on ranks that ran ahead;
  * integer counters (sums);
  * at the end of ``sample()``: the dwell times (to form the global time-major cumsum of
    markov_jump_hmc.py:321-324) and ONE all-gather of the sample columns.

Two communicators with the same small interface:

  ``RcclComm``  the product path: the library's own RCCL communicator (``mjhmc_comm_*`` of include/mjhmc_hip.h,
                librccl loaded with dlopen): the sample all-gather runs device ring -> device ring over xGMI and
                is re-tiled on the receiving GPU; no torch anywhere.
  ``Comm``      a torch.distributed shim kept for the CPU tests (gloo, world size 2: tests/test_parallel_gloo.py) and
                for running two ranks on ONE GPU (RCCL refuses two ranks on one device); host-staged.
"""
import ctypes
import os
import time

import numpy as np


class ShardPlan(object):
    """Contiguous column blocks: rank r owns [offset[r], offset[r] + count[r])."""

    def __init__(self, n_total, world):
        if int(n_total) < int(world):
            raise ValueError('cannot shard %d particle columns over %d ranks: every rank needs at least one'
                             % (n_total, world))
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


class _CommBase(object):
    on_device = False          # True: sample blocks are gathered ring-to-ring on the GPUs (RcclComm)

    def allreduce_sizes(self, m):
        v = np.zeros(self.world, dtype=np.int64)
        v[self.rank] = m
        return self.allreduce_ints(v, 'sum')


class FileRendezvous(object):
    """Host-side key/value exchange between the ranks of one job through a directory all of them see (a fresh
    ``mkdtemp`` directory made by the launcher: nothing stale can be in it).  Writes are atomic (tmp + rename)."""

    def __init__(self, dirpath, rank, world, timeout=300.0):
        self.dir, self.rank, self.world, self.timeout = str(dirpath), int(rank), int(world), float(timeout)

    def _path(self, key):
        return os.path.join(self.dir, key)

    def put(self, key, data):
        tmp = '%s.%d.tmp' % (self._path(key), os.getpid())
        with open(tmp, 'wb') as f:
            f.write(bytes(data))
        os.replace(tmp, self._path(key))

    def get(self, key):
        t0 = time.time()
        while not os.path.exists(self._path(key)):
            if time.time() - t0 > self.timeout:
                raise RuntimeError('rank %d: nothing at %s after %.0f s' % (self.rank, self._path(key), self.timeout))
            time.sleep(0.005)
        with open(self._path(key), 'rb') as f:
            return f.read()

    def all_agree(self, name, ok):
        """every rank posts a flag; True only when EVERY rank posted ok (decided before any collective is entered)"""
        self.put('%s.%d' % (name, self.rank), b'1' if ok else b'0')
        return all(self.get('%s.%d' % (name, r)) == b'1' for r in range(self.world))


def _process_start_ticks(pid):
    """start time of a process in clock ticks since boot (field 22 of /proc/<pid>/stat): (pid, start time) names one
    process for the lifetime of the machine, a recycled pid has another start time"""
    try:
        with open('/proc/%d/stat' % pid) as f:
            return f.read().rsplit(')', 1)[1].split()[19]
    except Exception:
        return '0'


def default_id_path():
    """Where rank 0 publishes the communicator id when the caller names no channel: ``MJHMC_COMM_ID_FILE`` (what
    ``bench.py --gpus N`` and any launcher that wants an explicit rendezvous export: a path in a fresh directory);
    last resort: a name built from the launcher's pid AND its start time plus MASTER_PORT, which every rank of one
    ``torch.distributed.run`` / ``mpirun`` job on a node shares and no earlier (crashed) job can have used."""
    path = os.environ.get('MJHMC_COMM_ID_FILE')
    if path:
        return path, True
    import tempfile
    ppid = os.getppid()
    return os.path.join(tempfile.gettempdir(), 'mjhmc_comm_%d_%s_%s.id'
                        % (ppid, _process_start_ticks(ppid), os.environ.get('MASTER_PORT', '0'))), False


class RcclComm(_CommBase):
    """This rank's end of the library's RCCL communicator.

    Rendezvous: rank 0 draws the 128-byte unique id and hands it to the other ranks through ``rendezvous`` (any object
    with ``put(key, bytes)`` / ``get(key) -> bytes``: FileRendezvous, or the launcher's key/value store), else through
    the file ``id_path`` (default: default_id_path())."""

    on_device = True
    _created = 0

    def __init__(self, rank=None, world=None, device=None, id_path=None, timeout=300.0, rendezvous=None):
        from . import _lib, engine
        self._lib_mod = _lib
        rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        device = int(os.environ.get('LOCAL_RANK', str(rank))) if device is None else int(device)
        self.rank, self.world, self.backend, self.device = rank, world, 'rccl', device
        self.ctx = engine.context(device)
        self.lib = self.ctx.lib
        serial = RcclComm._created                          # several communicators of one job: one id each
        RcclComm._created += 1
        if rendezvous is None:
            if id_path is None:
                id_path, _ = default_id_path()
            d, base = os.path.split('%s.%d' % (id_path, serial) if serial else id_path)
            rendezvous, key = FileRendezvous(d or '.', rank, world, timeout), base
        else:
            key = 'mjhmc_comm_id.%d' % serial
        buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        if rank == 0:
            _lib.check(self.lib.mjhmc_comm_unique_id(buf))
            rendezvous.put(key, buf.raw)
        else:
            raw = rendezvous.get(key)
            if len(raw) != _lib.COMM_ID_BYTES:
                raise RuntimeError('rank %d: communicator id of %d bytes (expected %d)' % (rank, len(raw), _lib.COMM_ID_BYTES))
            buf.raw = raw
        h = ctypes.c_void_p()
        _lib.check(self.lib.mjhmc_comm_create(self.ctx.handle, rank, world, buf, ctypes.byref(h)))
        self.handle = h
        self.barrier()                                   # everyone has read the id: rank 0 may remove the file
        if rank == 0 and isinstance(rendezvous, FileRendezvous):
            try:
                os.remove(os.path.join(rendezvous.dir, key))
            except OSError:
                pass

    def count(self):
        """ranks of the communicator as RCCL reports them (ncclCommCount)"""
        n = ctypes.c_int()
        self._lib_mod.check(self.lib.mjhmc_comm_count(self.handle, ctypes.byref(n)))
        return int(n.value)

    # -- small host-value collectives ---------------------------------------------------------
    def allreduce_ints(self, values, op='sum'):
        v = np.ascontiguousarray(np.asarray(values, dtype=np.int64).ravel()).copy()
        code = {'sum': self._lib_mod.OP_SUM, 'min': self._lib_mod.OP_MIN, 'max': self._lib_mod.OP_MAX}[op]
        self._lib_mod.check(self.lib.mjhmc_comm_allreduce_i64(self.handle, self._lib_mod.ptr(v), v.size, code))
        return v

    def allreduce_f64(self, values, op='sum'):
        v = np.ascontiguousarray(np.asarray(values, dtype=np.float64).ravel()).copy()
        code = {'sum': self._lib_mod.OP_SUM, 'min': self._lib_mod.OP_MIN, 'max': self._lib_mod.OP_MAX}[op]
        self._lib_mod.check(self.lib.mjhmc_comm_allreduce_f64(self.handle, self._lib_mod.ptr(v), v.size, code))
        return v.reshape(np.shape(values))

    def bcast(self, arr, src=0):
        v = np.ascontiguousarray(np.asarray(arr, dtype=np.float64)).copy()
        self._lib_mod.check(self.lib.mjhmc_comm_bcast(self.handle, self._lib_mod.ptr(v), v.nbytes, int(src)))
        return v

    def allgather_columns(self, block):
        """block (D, m_r) float64 with rank-dependent m_r -> list of the world's blocks (host values: state reads,
        dwelling times; the sample blocks take allgather_ring / allgather_picked instead)."""
        block = np.ascontiguousarray(block, dtype=np.float64)
        D = block.shape[0]
        sizes = self.allreduce_sizes(block.shape[1])
        nbytes = np.ascontiguousarray(sizes * D * 8, dtype=np.int64)
        out = np.empty(int(sizes.sum()) * D, dtype=np.float64)
        self._lib_mod.check(self.lib.mjhmc_comm_allgatherv(self.handle, self._lib_mod.ptr(block), self._lib_mod.ptr(nbytes),
                                                           self._lib_mod.ptr(out)))
        parts, at = [], 0
        for r in range(self.world):
            m = int(sizes[r])
            parts.append(out[at:at + D * m].reshape(D, m))
            at += D * m
        return parts

    def barrier(self):
        self.allreduce_ints([1], 'sum')

    # -- the data-path collective: sample columns, device ring to device ring --------------------
    def allgather_ring(self, dev, slot0, n, stacked, counts, root=None):
        """ring slots [slot0, slot0 + n) of every rank's DeviceSampler -> the unsharded sample block,
        (D, n * N_total) time-major or (D, N_total, n) when ``stacked``.  ``root``: only that rank re-tiles and downloads
        the block (the others take part in the collective and return None)."""
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        total = int(counts.sum())
        keep = root is None or int(root) == self.rank
        out = np.empty((dev.ndims, total, n) if stacked else (dev.ndims, n * total)) if keep else None
        self._lib_mod.check(self.lib.mjhmc_comm_allgather_ring(self.handle, dev.handle, int(slot0), int(n), 1 if stacked else 0,
                                                               self._lib_mod.ptr(counts), self._lib_mod.ptr(out)))
        return out

    def allgather_picked(self, dev, local_idx, counts):
        """every rank's picked ring columns (local pool indices), all-gathered: (D, sum counts), rank 0's first."""
        local_idx = np.ascontiguousarray(local_idx, dtype=np.int64)
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        out = np.empty((dev.ndims, int(counts.sum())))
        self._lib_mod.check(self.lib.mjhmc_comm_allgather_columns(self.handle, dev.handle, self._lib_mod.ptr(local_idx),
                                                                  local_idx.size, self._lib_mod.ptr(counts),
                                                                  self._lib_mod.ptr(out)))
        return out

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.mjhmc_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm(_CommBase):
    """torch.distributed shim (gloo on CPU for the tests; two ranks on one GPU): host-staged collectives.
    torch is imported here and nowhere else in the package."""

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

    def allreduce_f64(self, values, op='sum'):
        t = self._t(np.asarray(values, dtype=np.float64), self.torch.float64)
        ops = {'sum': self.dist.ReduceOp.SUM, 'min': self.dist.ReduceOp.MIN, 'max': self.dist.ReduceOp.MAX}
        self.dist.all_reduce(t, op=ops[op], group=self.group)
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
    """HMCBase.sample / resample=False (markov_jump_hmc.py:166-173,331-338) from per-rank host blocks (the
    host-staged path of the torch shim; RcclComm gathers the device rings instead, see allgather_ring).
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


def assemble_resample(comm, plan, n_samples, dwell_local, gather_local, uniforms=None, dev=None):
    """Dwell-time-weighted resampling of ContinuousTimeHMC.sample (markov_jump_hmc.py:309-329) over
    sharded columns.

    dwell_local : (n_samples, N_local) dwelling times recorded by this rank
    gather_local: f(local pool indices t * N_local + c) -> (D, m) columns from this rank's sample ring
                  (host-staged path), or None with ``dev`` = the rank's DeviceSampler when ``comm.on_device``
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
    owner = plan.owner_of(col_of)
    mine = np.nonzero(owner == comm.rank)[0]
    local_idx = t_of[mine] * n_loc + (col_of[mine] - lo)
    if comm.on_device and dev is not None:                                                      # the one sample gather
        counts = np.bincount(owner, minlength=comm.world).astype(np.int64)
        flat = comm.allgather_picked(dev, local_idx, counts)       # rank 0's columns, rank 1's, ...
        out = np.empty_like(flat)
        out[:, np.argsort(owner, kind='stable')] = flat            # back into sample order
        return out, sample_idx
    block = gather_local(local_idx)
    parts = comm.allgather_columns(block)
    out = np.empty((block.shape[0], sample_idx.size))
    for r, part in enumerate(parts):
        out[:, np.nonzero(owner == r)[0]] = part
    return out, sample_idx
