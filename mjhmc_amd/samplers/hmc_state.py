"""Particle-state views (mirrors mjhmc/samplers/hmc_state.py).

The live state lives in HBM (particle-major X, V plus per-particle EX, EV, H_flf, cache flag).
``DeviceHMCState`` materialises reference-shaped NumPy arrays on access; ``HMCState`` is a plain
host snapshot with the reference's field names, also accepted when assigning ``sampler.state``
(mjhmc/figures/poe_fig.py:59).
"""
import numpy as np

from .. import _lib


class HMCState(object):
    """Host snapshot: X, V, dEdX (ndims, nbatch); EX, EV (1, nbatch) (hmc_state.py:13-44)."""

    def __init__(self, X, parent=None, V=None, EX=None, EV=None, dEdX=None, cache_active=None):
        self.parent = parent
        self.X = X
        self.V = V
        self.nbatch = X.shape[1]
        self.active_idx = np.arange(self.nbatch)
        self.EX, self.EV, self.dEdX = EX, EV, dEdX
        self.cache_active = cache_active

    def H(self):
        return self.EX + self.EV

    def copy(self):
        c = lambda a: None if a is None else a.copy()
        return HMCState(self.X.copy(), self.parent, c(self.V), c(self.EX), c(self.EV), c(self.dEdX), c(self.cache_active))

    def get_state(self):
        return np.concatenate((self.X, self.V))


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
