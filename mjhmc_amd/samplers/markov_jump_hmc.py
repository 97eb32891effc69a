"""Sampler classes of the drop-in (mirrors mjhmc/samplers/markov_jump_hmc.py).

Same constructor keywords, methods and attributes as the reference; the per-iteration work
(``state.copy().L()/FLF()/R()``, ``transition_rates``, ``draw_from``, ``min_idx``,
``state.update`` -- markov_jump_hmc.py:355-415) is ONE fused HIP kernel launch per iteration on
the MI355X.  Extra keyword-only arguments: ``seed`` (counter-RNG seed; default drawn from
``np.random`` so ``np.random.seed`` still makes runs reproducible), ``dtype``, ``device``.
"""
from __future__ import print_function

import numpy as np

from .. import _lib
from .. import engine
from ..misc.distributions import Distribution
from .hmc_state import DeviceHMCState, HMCState

MAX_RETRY_DEPTH = 60


class HMCBase(object):
    """Hyper-parameters, counters and plumbing shared by all samplers (markov_jump_hmc.py:16-104)."""

    _mode = _lib.MODE_CONTROL

    def __init__(self, Xinit=None, E=None, dEdX=None, epsilon=1e-4, alpha=0.2, beta=None,
                 num_leapfrog_steps=5, distribution=None, seed=None, dtype=None, device=0, Vinit=None,
                 comm=None):
        self.num_leapfrog_steps = num_leapfrog_steps
        self.epsilon = epsilon
        self.beta = beta or alpha ** (1. / (self.epsilon * self.num_leapfrog_steps))
        self.original_epsilon = epsilon
        self.original_l = self.num_leapfrog_steps
        self.n_burn_in = 500
        self.p_flip = 0.5
        self.p_r = 1
        self.l_count = 0
        self.f_count = 0
        self.fl_count = 0
        self.r_count = 0
        self.grad_per_sample_step = self.num_leapfrog_steps
        self._seed, self._dtype, self._device, self._Vinit = seed, dtype, device, Vinit
        self._comm, self._plan = comm, None      # mjhmc_amd.parallel.Comm: shard particle columns over ranks
        self._pending = np.zeros(6, dtype=np.int64)  # local l, f, r, fl, E, dEdX increments not yet reduced
        self._acc_evals = np.zeros(2, dtype=np.int64)   # E / dEdX evaluations of the iteration in flight (retries included)
        self._iter_evals = []                           # per committed iteration of the CURRENT batch call: (E, dEdX), local
        self._dev = None
        self._in_retry = False
        if not isinstance(self, ContinuousTimeHMC):
            if not isinstance(distribution, Distribution):
                raise NotImplementedError(
                    'The MI355X engine runs energies that have a device functor: pass '
                    'distribution=<mjhmc_amd.misc.distributions.Distribution> (Xinit/E/dEdX callables '
                    'cannot be compiled for the GPU).')
            distribution.mjhmc = False
            distribution.reset()
            self._attach(distribution)

    # -- device plumbing -------------------------------------------------------------------
    def _attach(self, distribution):
        self.ndims = distribution.Xinit.shape[0]
        self.nbatch = distribution.Xinit.shape[1]
        self.energy_func = distribution.E
        self.grad_func = distribution.dEdX
        self.distribution = distribution
        seed = self._seed
        if seed is None:
            lo, hi = np.random.randint(0, 2 ** 32, size=2, dtype=np.uint64)
            seed = int(lo) | (int(hi) << 32)
            if self._comm is not None:                      # every rank must use rank 0's seed
                seed = int(self._comm.allreduce_ints([seed >> 1 if self._comm.rank == 0 else 0], 'sum')[0]) << 1
        self.seed = int(seed)
        X0, V0, first = distribution.Xinit, self._Vinit, 0
        if self._comm is not None:
            from ..parallel import ShardPlan
            self._plan = ShardPlan(self.nbatch, self._comm.world)
            first, stop = self._plan.span(self._comm.rank)
            X0 = X0[:, first:stop]
            V0 = None if V0 is None else V0[:, first:stop]
        self._dev = engine.make_sampler(distribution.bind(self._device), np.ascontiguousarray(X0), Vinit=V0,
                                        seed=self.seed, first_particle_id=first,
                                        dtype=self._dtype or getattr(distribution, 'state_dtype', 'float64'),
                                        mode=self._mode)
        self._dev.set_timing(False)     # (nothing here reads the device-side timing: two marker packets less per call)
        # HMCState.__init__ evaluates E and dEdX once on every particle (hmc_state.py:28-39)
        distribution.E_count += self.nbatch
        distribution.dEdX_count += self.nbatch

    def _push_hparams(self):
        self._dev.set_hparams(self.epsilon, self.num_leapfrog_steps, self.p_r, self.beta, self.p_flip)

    @property
    def state(self):
        return DeviceHMCState(self)

    @state.setter
    def state(self, Z):
        """Assigning an HMCState uploads its X (and V) (figures/poe_fig.py:59); with sharded columns every rank
        uploads its own block."""
        cols = slice(None) if self._plan is None else slice(*self._plan.span(self._comm.rank))
        self._dev.write(_lib.F_X, np.ascontiguousarray(Z.X[:, cols]))
        if getattr(Z, 'V', None) is not None:
            self._dev.write(_lib.F_V, np.ascontiguousarray(Z.V[:, cols]))

    # -- checkpoint / resume -----------------------------------------------------------------
    _SAVED = ('epsilon', 'num_leapfrog_steps', 'beta', 'p_r', 'p_flip', 'original_epsilon', 'original_l',
              'l_count', 'f_count', 'fl_count', 'r_count')

    def save_state(self, path):
        """Everything a run needs to continue exactly where it stands -- X, V, the inverse-L cache (H_flf, NaN =
        cold), the counter-RNG position, hyper-parameters and counters -- as an ``.npz`` file.  A sampler built on
        the same distribution and ``load_state``-ed continues bit for bit (the RNG is a pure function of
        (seed, particle id, tick))."""
        st = self.state
        meta = {k: np.asarray(getattr(self, k)) for k in self._SAVED}
        np.savez(path, X=st.X, V=st.V, H_flf=st.H_flf[0], dwelling_times=np.asarray(getattr(self, 'dwelling_times', 0.0)),
                 tick=np.uint64(self._dev.get_tick()), seed=np.uint64(self.seed),
                 E_count=self.distribution.E_count, dEdX_count=self.distribution.dEdX_count, **meta)

    def load_state(self, path):
        z = np.load(path if str(path).endswith('.npz') else str(path) + '.npz')
        if z['X'].shape != (self.ndims, self.nbatch):
            raise ValueError('checkpoint holds a %r state, this sampler is %r' % (z['X'].shape, (self.ndims, self.nbatch)))
        if int(z['seed']) != self.seed:
            raise ValueError('checkpoint was written with seed %d, this sampler has %d' % (int(z['seed']), self.seed))
        cols = slice(None) if self._plan is None else slice(*self._plan.span(self._comm.rank))
        self._dev.write(_lib.F_X, np.ascontiguousarray(z['X'][:, cols]))      # re-derives EX (and dE/dX), clears the cache
        self._dev.write(_lib.F_V, np.ascontiguousarray(z['V'][:, cols]))
        self._dev.write(_lib.F_HFLF, np.ascontiguousarray(z['H_flf'][cols]))
        self._dev.set_tick(int(z['tick']))
        for k in self._SAVED:
            setattr(self, k, z[k].item())
        self.distribution.E_count, self.distribution.dEdX_count = int(z['E_count']), int(z['dEdX_count'])
        if hasattr(self, 'dwelling_times'):
            self.dwelling_times = z['dwelling_times']
        return self

    def E(self, X):
        return self.energy_func(X).reshape((1, -1))

    def dEdX(self, X):
        return self.grad_func(X)

    def leap_prob(self, Z1, Z2):
        """Metropolis-Hastings probability of transitioning from state Z1 to state Z2 (markov_jump_hmc.py:106-114)."""
        Ediff = Z1.H() - Z2.H()
        p_acc = np.ones((1, Ediff.shape[1]))
        p_acc[Ediff < 0] = np.exp(Ediff[Ediff < 0])
        return p_acc

    def burn_in(self):
        self._run(self.n_burn_in)
        self._publish()

    def _account(self, st):
        self._pending[4] += st.E_evals
        self._pending[5] += st.dEdX_evals
        self._acc_evals += (st.E_evals, st.dEdX_evals)

    def _commit(self, st):
        self._pending[:4] += (st.l, st.f, st.r, st.fl)
        self._iter_evals.append(self._acc_evals.copy())
        self._acc_evals[:] = 0

    def eval_trace(self, n_last):
        """(E_evals, dEdX_evals) per iteration for the last ``n_last`` committed iterations of the most recent
        batch call (sample / _record / burn_in), summed over ranks: the increments of Distribution.E_count /
        dEdX_count a per-step host loop would observe (mjhmc/misc/autocor.py:246-248)."""
        tr = np.array(self._iter_evals[-n_last:], dtype=np.int64).reshape(-1, 2)
        if self._comm is not None:
            tr = self._comm.allreduce_ints(tr.ravel(), 'sum').reshape(-1, 2)
        return tr

    def _publish(self):
        """Fold this call's integer bookkeeping into the public counters (summed over ranks)."""
        inc = self._pending if self._comm is None else self._comm.allreduce_ints(self._pending, 'sum')
        self.l_count += int(inc[0])
        self.f_count += int(inc[1])
        self.r_count += int(inc[2])
        self.fl_count += int(inc[3])
        self.distribution.E_count += int(inc[4])
        self.distribution.dEdX_count += int(inc[5])
        self._pending[:] = 0

    # -- iteration driver (discrete-time samplers; the jump processes override _one) -----------
    def _one(self, ring_slot=-1, replay=None):
        self._push_hparams()
        rn = ru = None
        if replay is not None:
            rn, ru = replay.pop(0)
        stats, n_done = self._dev.iterate(1, replay_normal=rn, replay_unif=ru, ring_slot0=ring_slot)
        self._account(stats[0])
        self._commit(stats[0])

    def _run(self, n_iter, ring_slot0=-1, replay=None, download=None, keep_trace=False):
        """n_iter iterations launched back to back; the host only steps in on a non-finite rate.  ``download = (out, k0)``:
        ring slot ring_slot0 + i also goes to out[:, (k0 + i) * N : ...] while the following iterations run
        (mjhmc_iterate_download).  ``keep_trace``: this call continues a batch call that walks the device ring in chunks --
        eval_trace() describes the whole batch call, not its last chunk."""
        if not self._in_retry and not keep_trace:
            self._iter_evals = []                         # the trace describes one batch call, it does not grow for ever
        if replay is not None:                            # recorded random numbers: one attempt at a time
            for i in range(n_iter):
                self._one(ring_slot0 + i if ring_slot0 >= 0 else -1, replay)
            return
        # only the jump processes can meet a non-finite rate; the discrete-time samplers never roll back
        sync = self._comm is not None and self._mode != _lib.MODE_CONTROL
        done = 0
        while done < n_iter:
            self._push_hparams()
            slot = ring_slot0 + done if ring_slot0 >= 0 else -1
            todo = n_iter - done
            if sync and todo > 1:
                self._dev.checkpoint()                    # once per batch; a single iteration needs none (rollback)
            if download is not None:
                stats, n_done = self._dev.iterate_download(todo, slot, download[0], download[1] + done)
            else:
                stats, n_done = self._dev.iterate(todo, ring_slot0=slot)
            if sync:
                # the reference aborts the WHOLE batch on one bad particle: every rank keeps only the
                # iterations all ranks committed; a rank that ran ahead rolls back + replays (bit-identical:
                # the RNG is a pure function of (seed, particle id, tick))
                from ..parallel import agree_on_progress
                common = agree_on_progress(self._comm, n_done)
                if common < n_done:
                    if todo == 1:
                        self._dev.rollback()              # ping-pong inputs are intact; the tick stays consumed
                    else:
                        self._dev.restore()
                        if common:
                            redo, again = self._dev.iterate(common, ring_slot0=slot)
                            assert again == common
                        self._dev.advance_tick(1)         # the failed attempt's tick, consumed everywhere
                failed_somewhere = common < todo
                n_done = common
            else:
                failed_somewhere = len(stats) > n_done
            for st in stats[:n_done]:
                self._account(st)
                self._commit(st)
            done += n_done
            if failed_somewhere:                          # the attempt after the committed ones failed
                self._account(stats[n_done])
                self._retry(ring_slot0 + done if ring_slot0 >= 0 else -1, None)
                if download is not None:                  # the retried iteration's slot, plainly
                    N = self._dev.nparticles
                    k = download[1] + done
                    download[0][:, k * N:(k + 1) * N] = self._dev.ring_read(ring_slot0 + done, 1)
                done += 1

    def _retry(self, ring_slot, replay):
        raise ValueError('Infinite rate.')                # only the jump processes can get here

    def sampling_iteration(self, replay=None):
        """One step of every particle (markov_jump_hmc.py:116-148).  ``replay=[(normals (D,N),
        uniforms (2N+1) = accept, flip, R gate), ...]`` feeds recorded random numbers."""
        self._step(replay)
        self._publish()

    def _step(self, replay):
        if self._comm is not None:
            self._run(1)
        else:
            self._iter_evals = []
            self._one(-1, replay)

    # Sharded runs (extension): ``sampler.gather_root = r`` makes sample() with resample=False return the gathered block on
    # rank r only -- the other ranks take part in the collective, skip the re-tile and the download of a block they do not
    # want (mjhmc_comm_allgather_ring with host_out = NULL) and get None.  Default None: every rank gets the block, as a
    # single-process caller of the reference does.
    gather_root = None

    def _stack(self, n_samples, preserve_order, out=None):
        if self._comm is not None and self._comm.on_device:
            # the one data-path collective: device rings all-gathered over RCCL, re-tiled on the receiving GPU
            res = self._comm.allgather_ring(self._dev, 0, n_samples, bool(preserve_order), self._plan.counts, root=self.gather_root)
        else:
            local = self._dev.ring_read(0, n_samples, stacked=bool(preserve_order), out=out if self._comm is None else None)
            if self._comm is None:
                return local
            from ..parallel import assemble_stacked
            res = assemble_stacked(self._comm, self._plan, local, n_samples, bool(preserve_order))
            if self.gather_root is not None and int(self.gather_root) != self._comm.rank:
                res = None
        if res is None:
            return None
        if out is None:
            return res
        if out.shape != res.shape or out.dtype != np.float64:      # sharded run: the gathered block lands in the caller's array
            raise ValueError('out must be a float64 array of shape %r' % (res.shape,))
        out[...] = res
        return out

    def sample(self, n_samples=1000, preserve_order=False, replay=None, out=None):
        """markov_jump_hmc.py:150-173.  ``out`` (extension): a preallocated C-contiguous float64 array of the result's
        shape, (ndims, n_samples * nbatch) or (ndims, nbatch, n_samples), filled and returned instead of a fresh one."""
        if self._streams(preserve_order, replay):
            return self._sample_streamed(n_samples, out)
        self._record(n_samples, replay)
        return self._stack(n_samples, preserve_order, out)

    def _streams(self, preserve_order, replay):
        """np.concatenate(samples, axis=1) of an unsharded run with the counter RNG: every sample goes to the host while the
        next iterations run, and the device ring may be smaller than the run"""
        return self._comm is None and replay is None and not preserve_order and hasattr(self._dev, 'iterate_download')

    def _host_array(self, n_states, what, out=None):
        shape = (self.ndims, n_states * self.nbatch)
        if out is not None:
            if out.shape != shape or out.dtype != np.float64 or not out.flags.c_contiguous:
                raise ValueError('out must be a C-contiguous float64 array of shape %r' % (shape,))
            return out
        try:
            return np.empty(shape)
        except MemoryError:
            raise MemoryError('%s: %d states of %d x %d float64 are %.1f GB of host memory, which this machine does not give -- '
                              'draw fewer samples per call, or thin them' % (what, n_states, self.ndims, self.nbatch,
                                                                             8e-9 * self.ndims * self.nbatch * n_states))

    def _sample_streamed(self, n_iter, out=None, what='sample()'):
        """n_iter iterations, the state after each as columns [k N, (k + 1) N) of the returned (ndims, n_iter * nbatch) array.
        The device ring holds as many slots as fit (all of them if it can): a run bigger than the device walks it in
        chunks."""
        out = self._host_array(n_iter, what, out)
        slots = max(2, self._dev.ring_budget_slots(n_iter))
        self._dev.ring_alloc(slots)
        done = 0
        while done < n_iter:
            chunk = min(slots, n_iter - done)
            self._run(chunk, ring_slot0=0, download=(out, done), keep_trace=done > 0)
            done += chunk
        self._publish()
        return out

    def _record(self, n_samples, replay=None):
        """Run n_samples iterations, snapshotting X after each into device ring slots [0, n_samples)."""
        self._dev.ring_alloc(n_samples)
        self._run(n_samples, ring_slot0=0, replay=replay)
        self._publish()


class HMC(HMCBase):
    def __init__(self, *args, **kwargs):
        super(HMC, self).__init__(*args, **kwargs)
        self.p_flip = 1


class ControlHMC(HMCBase):
    def __init__(self, *args, **kwargs):
        super(ControlHMC, self).__init__(*args, **kwargs)
        self.p_flip = 1
        self.p_r = - np.log(1 - self.beta) * 0.5
        self.beta = 1


class ContinuousTimeHMC(HMCBase):
    """Base of the jump-process samplers (markov_jump_hmc.py:203-347)."""

    _mode = _lib.MODE_CTHMC

    def __init__(self, *args, **kwargs):
        self.resample = kwargs.pop('resample', True)
        distribution = kwargs.get('distribution')
        super(ContinuousTimeHMC, self).__init__(*args, **kwargs)
        if not (0 <= self.beta < 1):
            # the reference would recurse forever: p_r = -log(1-beta)/2 is inf/nan and every
            # attempt raises in draw_from (markov_jump_hmc.py:221, 376-385)
            raise ValueError('beta must satisfy 0 <= beta < 1, got %r' % (self.beta,))
        self.p_r = - np.log(1 - self.beta) * 0.5
        self.beta = 1
        if isinstance(distribution, Distribution):
            distribution.mjhmc = True
            if not distribution.generation_instance:
                distribution.reset()
            self._attach(distribution)
        else:
            raise NotImplementedError(
                'Unfortunately, you must define your distribution by subclassing '
                'mjhmc_amd.misc.distributions.Distribution (same rule as the reference, '
                'markov_jump_hmc.py:235-242).')
        self.dwelling_times = np.zeros(self.nbatch)

    def transition_rates(self, Z1, Z2):
        Ediff = Z1.H() - Z2.H()
        return np.exp(Ediff) ** .5

    # -- iteration driver --------------------------------------------------------------------
    def _retry(self, ring_slot, replay):
        """ContinuousTimeHMC lets draw_from's ValueError propagate (markov_jump_hmc.py:266-268)."""
        raise ValueError("Infinite rate. This occurs when calculating transition rates between states that "
                         "have a very large energy difference (mjhmc/misc/utils.py:43-48).")

    def _one(self, ring_slot=-1, replay=None):
        """One sampling_iteration including the reference's retry recursion."""
        self._push_hparams()
        rn = re = None
        if replay is not None:
            rn, re = replay.pop(0)
        stats, n_done = self._dev.iterate(1, replay_normal=rn, replay_exp=re, ring_slot0=ring_slot)
        self._account(stats[0])
        if n_done == 1:
            self._commit(stats[0])
        else:
            self._retry(ring_slot, replay)

    def _read_dwell(self):
        d = self._dev.read(_lib.F_DWELL)
        if self._comm is not None:
            from ..parallel import gather_vector
            d = gather_vector(self._comm, self._plan, d)
        self.dwelling_times = d

    def sampling_iteration(self, replay=None):
        """One jump of every particle.  ``replay=[(normals (D,N), unit_exps (3,N)), ...]`` feeds
        recorded random numbers (one pair per attempt) instead of the counter RNG."""
        self._step(replay)
        self._publish()
        self._read_dwell()

    def burn_in(self):
        super(ContinuousTimeHMC, self).burn_in()
        self._read_dwell()

    def sample(self, n_samples=1000, preserve_order=False, num_steps=None, replay=None, out=None):
        """markov_jump_hmc.py:293-338.  ``num_steps`` is accepted as an alias of ``n_samples``
        (the README calls ``sample(num_steps=10)``, README.md:36).  ``out`` (extension, ``resample=False`` only): a
        preallocated array to fill, see HMCBase.sample."""
        if num_steps is not None:
            n_samples = num_steps
        if self.resample:
            # (this path records into the ring and gathers columns from it: no staging copy per slot is charged)
            if self._streams(False, replay) and self._dev.ring_budget_slots(n_samples + 1, staging=False) < n_samples + 1:
                return self._resample_on_host(n_samples)
            self._dev.ring_alloc(n_samples + 1)
            self._run(n_samples + 1, ring_slot0=0, replay=replay)
            self._publish()
            self._read_dwell()
            if self._comm is not None:
                from ..parallel import assemble_resample
                out, self._last_resample_idx = assemble_resample(
                    self._comm, self._plan, n_samples, self._dev.ring_read_dwell(0, n_samples), self._dev.ring_gather,
                    dev=self._dev)
                return out
            dwell_t = self._dev.ring_read_dwell(0, n_samples).reshape(-1)   # time-major, as np.concatenate
            total_t = np.sum(dwell_t)
            cumul_t = np.cumsum(dwell_t)
            rand_vals = np.sort(np.random.random(n_samples * self.nbatch)) * total_t
            # first index with cumul_t > r, for every r at once (the reference loops, :326-328)
            sample_idx = np.searchsorted(cumul_t, rand_vals, side='right')
            if sample_idx.size and sample_idx[-1] >= dwell_t.size:
                raise IndexError('index 0 is out of bounds for axis 0 with size 0')   # infinite dwell time
            self._last_resample_idx = sample_idx
            return self._dev.ring_gather(sample_idx)
        if self._streams(preserve_order, replay):
            res = self._sample_streamed(n_samples, out)
            self._read_dwell()
            return res
        self._record(n_samples, replay)
        return self._stack(n_samples, preserve_order, out)

    def _resample_on_host(self, n_samples):
        """sample() with dwell-time resampling when the n_samples + 1 states do not fit the device: they are streamed to
        the host through a ring of the slots that do fit, and the columns are picked there (same uniforms, same indices)."""
        n_iter = n_samples + 1
        states = self._host_array(n_iter, 'sample(n_samples=%d, resample=True)' % n_samples)
        slots = max(2, self._dev.ring_budget_slots(n_iter))
        self._dev.ring_alloc(slots)
        dwell, done = [], 0
        while done < n_iter:
            chunk = min(slots, n_iter - done)
            self._run(chunk, ring_slot0=0, download=(states, done), keep_trace=done > 0)
            dwell.append(self._dev.ring_read_dwell(0, chunk))
            done += chunk
        self._publish()
        self._read_dwell()
        dwell_t = np.concatenate(dwell)[:n_samples].reshape(-1)
        cumul_t = np.cumsum(dwell_t)
        rand_vals = np.sort(np.random.random(n_samples * self.nbatch)) * np.sum(dwell_t)
        sample_idx = np.searchsorted(cumul_t, rand_vals, side='right')
        if sample_idx.size and sample_idx[-1] >= dwell_t.size:
            raise IndexError('index 0 is out of bounds for axis 0 with size 0')   # infinite dwell time
        self._last_resample_idx = sample_idx
        return states[:, sample_idx]

    def _record(self, n_samples, replay=None):
        super(ContinuousTimeHMC, self)._record(n_samples, replay)
        self._read_dwell()


class MarkovJumpHMC(ContinuousTimeHMC):
    """Markov Jump HMC, arXiv:1509.03808 (markov_jump_hmc.py:350-415)."""

    _mode = _lib.MODE_MJHMC

    def _retry(self, ring_slot, replay):
        """markov_jump_hmc.py:376-389: halve epsilon, double L, wipe the FLF cache, try again, restore."""
        self.epsilon *= 0.5
        self.num_leapfrog_steps *= 2
        depth = np.log(self.original_epsilon / self.epsilon) / np.log(2)
        print("Ecountered infinite rate, doubling back. Depth: {}".format(depth))
        if depth > MAX_RETRY_DEPTH:
            raise RuntimeError('non-finite transition rates persist after %d halvings' % MAX_RETRY_DEPTH)
        self._dev.reset_flf_cache()
        nested, self._in_retry = self._in_retry, True
        try:
            if self._comm is not None:
                self._run(1, ring_slot)
            else:
                self._one(ring_slot, replay)
        finally:
            self._in_retry = nested
        self.epsilon *= 2
        self.num_leapfrog_steps = int(self.num_leapfrog_steps / 2)
