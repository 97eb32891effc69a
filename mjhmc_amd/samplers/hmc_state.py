"""Particle-state views (mirrors mjhmc/samplers/hmc_state.py).

The live state lives in HBM (particle-major X, V plus per-particle EX, EV, H_flf, cache flag).
``DeviceHMCState`` materialises reference-shaped NumPy arrays on access; ``HMCState`` is a plain
host snapshot with the reference's field names, also accepted when assigning ``sampler.state``
(mjhmc/figures/poe_fig.py:59).
"""
import numpy as np

from .. import _lib


class HMCState(object):
    """Host snapshot: X, V, dEdX (ndims, nbatch); EX, EV (1, nbatch) (hmc_state.py:13-44), with the reference's
    state operators.  ``L`` / ``leapfrog`` integrate on the device (mjhmc_leapfrog, the reference's literal
    operation order); the others are bookkeeping on the host arrays, like the reference's.  Operators act on the
    columns in ``active_idx`` and return ``self``, as in the reference."""

    def __init__(self, X, parent=None, V=None, EX=None, EV=None, dEdX=None, cache_active=None):
        self.parent = parent
        self.X = X
        self.V = V
        self.nbatch = X.shape[1]
        self.active_idx = np.arange(self.nbatch)
        self.EX, self.EV, self.dEdX = EX, EV, dEdX
        self.cache_active = np.zeros(self.nbatch, dtype=bool) if cache_active is None else cache_active
        self.cached_flf_state = None

    def H(self):
        return self.EX + self.EV

    def copy(self, copy_slave=False):
        c = lambda a: None if a is None else a.copy()
        Z = HMCState(self.X.copy(), self.parent, c(self.V), c(self.EX), c(self.EV), c(self.dEdX), c(self.cache_active))
        Z.active_idx = self.active_idx.copy()
        if self.cached_flf_state is not None and not copy_slave:
            Z.cached_flf_state = self.cached_flf_state.copy(True)
        return Z

    def get_state(self):
        return np.concatenate((self.X, self.V))

    # -- hmc_state.py:46-53 -----------------------------------------------------------------------
    def update_EX(self):
        self.EX[:, self.active_idx] = self.parent.E(self.X[:, self.active_idx]).reshape((1, -1))

    def update_EV(self):
        self.EV[:, self.active_idx] = np.sum(self.V[:, self.active_idx] ** 2, axis=0).reshape((1, -1)) / 2.

    def update_dEdX(self):
        self.dEdX[:, self.active_idx] = self.parent.dEdX(self.X[:, self.active_idx])

    # -- hmc_state.py:62-72 -----------------------------------------------------------------------
    def update(self, idx, Z):
        """replace batch elements idx with state from Z"""
        if len(idx) == 0:
            return
        self.X[:, idx] = Z.X[:, idx]
        self.V[:, idx] = Z.V[:, idx]
        self.EX[:, idx] = Z.EX[:, idx]
        self.EV[:, idx] = Z.EV[:, idx]
        self.dEdX[:, idx] = Z.dEdX[:, idx]

    # -- hmc_state.py:86-129 ----------------------------------------------------------------------
    def _integrate(self, n_steps, energies):
        idx = self.active_idx
        if len(idx) == 0:
            return
        dist = self.parent.distribution
        X, V, EX, EV, G = dist.bind(self.parent._device).leapfrog(
            self.X[:, idx], self.V[:, idx], self.parent.epsilon, n_steps,
            dtype=self.parent._dtype or getattr(dist, 'state_dtype', 'float64'))
        self.X[:, idx], self.V[:, idx], self.dEdX[:, idx] = X, V, G
        if energies:
            self.EV[0, idx], self.EX[0, idx] = EV, EX
        dist.dEdX_count += n_steps * len(idx)                  # Distribution.E / dEdX count their calls (:62-75)
        if energies:
            dist.E_count += len(idx)

    def leapfrog(self):
        """A single leapfrog step for X and V (energies are not refreshed, as in the reference)."""
        self._integrate(1, False)

    def L(self):
        """Run the leapfrog operator for num_leapfrog_steps steps; returns self."""
        self._integrate(self.parent.num_leapfrog_steps, True)
        return self

    def F(self):
        self.V[:, self.active_idx] = - self.V[:, self.active_idx]
        return self

    def FLF(self):
        """F L F of the columns whose inverse-L cache is cold; cached columns are copied from the cached state when
        this snapshot carries one (a snapshot taken from the device keeps only that state's energy, H_flf, which is all
        the sampler ever reads of it: every column is then integrated)."""
        if self.cached_flf_state is None:
            return self.F().L().F()
        cached_idx = np.where(self.cache_active == True)[0]     # noqa: E712
        self.active_idx = np.where(self.cache_active == False)[0]   # noqa: E712
        flf_state = self.F().L().F()
        flf_state.update(cached_idx, self.cached_flf_state)
        self.active_idx = np.arange(self.nbatch)
        return flf_state

    def R(self):
        """randomizes the momentum with rate beta (process-global NumPy stream, like the reference)"""
        self.V = self.V * np.sqrt(1. - self.parent.beta) + np.random.randn(
            self.parent.ndims, self.nbatch) * np.sqrt(self.parent.beta)
        self.update_EV()
        return self

    # -- hmc_state.py:131-148 ---------------------------------------------------------------------
    def cache_flf_state(self, idx, Z):
        if self.cached_flf_state is None:
            self.cached_flf_state = self.copy(True)
        self.cached_flf_state.update(idx, Z)
        self.cache_active[idx] = True

    def clear_flf_cache(self, idx):
        self.cache_active[idx] = False

    def reset_flf_cache(self):
        self.cache_active = np.zeros_like(self.cache_active)


class DeviceHMCState(object):
    """Read-through view of the sampler's device state."""

    def __init__(self, parent):
        self.parent = parent
        self.nbatch = parent.nbatch
        self.active_idx = np.arange(self.nbatch)

    def _read(self, field):
        local = self.parent._dev.read(field)
        comm = getattr(self.parent, '_comm', None)
        if comm is None:
            return local
        from ..parallel import gather_state_columns
        if local.ndim == 2:
            return gather_state_columns(comm, self.parent._plan, local)
        return gather_state_columns(comm, self.parent._plan, local.astype(np.float64).reshape(1, -1))[0].astype(local.dtype)

    @property
    def X(self):
        return self._read(_lib.F_X)

    @property
    def V(self):
        return self._read(_lib.F_V)

    @property
    def EX(self):
        return self._read(_lib.F_EX).reshape((1, -1))

    @property
    def EV(self):
        return self._read(_lib.F_EV).reshape((1, -1))

    @property
    def dEdX(self):
        return self._read(_lib.F_DEDX)

    @property
    def cache_active(self):
        return self._read(_lib.F_CACHE).astype(bool)

    @property
    def H_flf(self):
        """H() of the cached inverse-L state (valid where cache_active)."""
        return self._read(_lib.F_HFLF).reshape((1, -1))

    def H(self):
        return self.EX + self.EV

    def copy(self):
        return HMCState(self.X, self.parent, self.V, self.EX, self.EV, self.dEdX, self.cache_active)

    def get_state(self):
        return np.concatenate((self.X, self.V))

    def reset_flf_cache(self):
        self.parent._dev.reset_flf_cache()

    # -- the reference's state operators on the LIVE state (hmc_state.py:62-148: there `sampler.state` is the HMCState
    #    itself, so `sampler.state.L()` moves the sampler).  Here: snapshot, operate, write X and V back -- the write
    #    re-derives EX / EV / dE/dX on the device and clears the inverse-L cache, as a move does.
    def _apply(self, op):
        Z = self.copy()
        op(Z)
        self.parent.state = Z
        return self

    def leapfrog(self):
        self._apply(lambda Z: Z.leapfrog())

    def L(self):
        return self._apply(lambda Z: Z.L())

    def F(self):
        return self._apply(lambda Z: Z.F())

    def FLF(self):
        return self._apply(lambda Z: Z.FLF())

    def R(self):
        return self._apply(lambda Z: Z.R())

    def update(self, idx, Z):
        """replace batch elements idx with state from Z"""
        idx = np.asarray(idx, dtype=np.int64)
        if idx.size:
            self._apply(lambda S: S.update(idx, Z))

    def _write_h_flf(self, h):
        cols = slice(None) if getattr(self.parent, '_plan', None) is None else slice(*self.parent._plan.span(self.parent._comm.rank))
        self.parent._dev.write(_lib.F_HFLF, np.ascontiguousarray(h[cols], dtype=np.float64))

    def cache_flf_state(self, idx, Z):
        """The device keeps H() of the cached inverse-L state, which is all the sampler ever reads of it
        (markov_jump_hmc.py:367): columns idx take Z's."""
        h = self.H_flf[0].copy()
        h[idx] = np.asarray(Z.H())[0, idx]
        self._write_h_flf(h)

    def clear_flf_cache(self, idx):
        h = self.H_flf[0].copy()
        h[idx] = np.nan
        self._write_h_flf(h)

    @property
    def cached_flf_state(self):
        """Not materialised on the device (only its energy is: ``H_flf``)."""
        return None
