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
                 num_leapfrog_steps=5, distribution=None, seed=None, dtype='float64', device=0, Vinit=None):
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
        self._dev = None
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
        self.seed = int(seed)
        self._dev = engine.DeviceSampler(distribution.bind(self._device), distribution.Xinit.copy(), Vinit=self._Vinit,
                                         seed=self.seed, dtype=self._dtype, mode=self._mode)
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
        """Assigning an HMCState uploads its X (and V) (figures/poe_fig.py:59)."""
        self._dev.write(_lib.F_X, Z.X)
        if getattr(Z, 'V', None) is not None:
            self._dev.write(_lib.F_V, Z.V)

    def E(self, X):
        return self.energy_func(X).reshape((1, -1))

    def dEdX(self, X):
        return self.grad_func(X)

    def burn_in(self):
        self._run(self.n_burn_in)

    def _account(self, st):
        self.distribution.E_count += st.E_evals
        self.distribution.dEdX_count += st.dEdX_evals

    def _commit(self, st):
        self.l_count += st.l
        self.f_count += st.f
        self.r_count += st.r
        self.fl_count += st.fl

    # -- iteration driver (discrete-time samplers; the jump processes override _one) -----------
    def _one(self, ring_slot=-1, replay=None):
        self._push_hparams()
        rn = ru = None
        if replay is not None:
            rn, ru = replay.pop(0)
        stats, n_done = self._dev.iterate(1, replay_normal=rn, replay_unif=ru, ring_slot0=ring_slot)
        self._account(stats[0])
        self._commit(stats[0])

    def _run(self, n_iter, ring_slot0=-1, replay=None):
        """n_iter iterations launched back to back; the host only steps in on a non-finite rate."""
        if replay is not None:                            # recorded random numbers: one attempt at a time
            for i in range(n_iter):
                self._one(ring_slot0 + i if ring_slot0 >= 0 else -1, replay)
            return
        done = 0
        while done < n_iter:
            self._push_hparams()
            slot = ring_slot0 + done if ring_slot0 >= 0 else -1
            stats, n_done = self._dev.iterate(n_iter - done, ring_slot0=slot)
            for st in stats[:n_done]:
                self._account(st)
                self._commit(st)
            done += n_done
            if len(stats) > n_done:                       # the attempt after the committed ones failed
                self._account(stats[n_done])
                self._retry(ring_slot0 + done if ring_slot0 >= 0 else -1, None)
                done += 1

    def _retry(self, ring_slot, replay):
        raise ValueError('Infinite rate.')                # only the jump processes can get here

    def sampling_iteration(self, replay=None):
        """One step of every particle (markov_jump_hmc.py:116-148).  ``replay=[(normals (D,N),
        uniforms (2N+1) = accept, flip, R gate), ...]`` feeds recorded random numbers."""
        self._one(-1, replay)

    def sample(self, n_samples=1000, preserve_order=False, replay=None):
        """markov_jump_hmc.py:150-173."""
        self._dev.ring_alloc(n_samples)
        self._run(n_samples, ring_slot0=0, replay=replay)
        return self._dev.ring_read(0, n_samples, stacked=bool(preserve_order))


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

    def _run(self, n_iter, ring_slot0=-1, replay=None):
        super(ContinuousTimeHMC, self)._run(n_iter, ring_slot0, replay)
        self.dwelling_times = self._dev.read(_lib.F_DWELL)

    def sampling_iteration(self, replay=None):
        """One jump of every particle.  ``replay=[(normals (D,N), unit_exps (3,N)), ...]`` feeds
        recorded random numbers (one pair per attempt) instead of the counter RNG."""
        self._one(-1, replay)
        self.dwelling_times = self._dev.read(_lib.F_DWELL)

    def sample(self, n_samples=1000, preserve_order=False, num_steps=None, replay=None):
        """markov_jump_hmc.py:293-338.  ``num_steps`` is accepted as an alias of ``n_samples``
        (the README calls ``sample(num_steps=10)``, README.md:36)."""
        if num_steps is not None:
            n_samples = num_steps
        if self.resample:
            self._dev.ring_alloc(n_samples + 1)
            self._run(n_samples + 1, ring_slot0=0, replay=replay)
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
        self._dev.ring_alloc(n_samples)
        self._run(n_samples, ring_slot0=0, replay=replay)
        return self._dev.ring_read(0, n_samples, stacked=bool(preserve_order))


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
        self._one(ring_slot, replay)
        self.epsilon *= 2
        self.num_leapfrog_steps = int(self.num_leapfrog_steps / 2)
