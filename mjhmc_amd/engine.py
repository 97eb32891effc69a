"""Object wrappers over the C ABI: Context (device), DeviceEnergy, DeviceSampler.

These are plumbing; the reference-shaped API lives in mjhmc_amd.samplers / mjhmc_amd.misc.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check, ptr, as_f64

_DTYPES = {'float64': _lib.F64, 'f64': _lib.F64, np.float64: _lib.F64, 'float32': _lib.F32, 'f32': _lib.F32,
           np.float32: _lib.F32, 'bfloat16': _lib.BF16, 'bf16': _lib.BF16}

_contexts = {}


def dtype_code(dtype):
    try:
        return _DTYPES[dtype]
    except KeyError:
        raise ValueError('dtype must be float64, float32 or bfloat16, got %r' % (dtype,))


class Context(object):
    def __init__(self, device=0, lib=None):
        self.lib = lib if lib is not None else _lib.load()     # lib: _lib.load_test_hooks() in the A/B tests
        h = ctypes.c_void_p()
        check(self.lib.mjhmc_ctx_create(int(device), ctypes.byref(h)), self.lib)
        self.handle = h
        self.device = int(device)

    def info(self):
        name = ctypes.create_string_buffer(256)
        ncu = ctypes.c_int()
        hbm = ctypes.c_uint64()
        check(self.lib.mjhmc_ctx_info(self.handle, name, 256, ctypes.byref(ncu), ctypes.byref(hbm)), self.lib)
        return dict(name=name.value.decode(), n_cu=ncu.value, hbm_bytes=hbm.value)

    def mem_info(self):
        """(free, total) device memory in bytes"""
        f, t = ctypes.c_uint64(), ctypes.c_uint64()
        check(self.lib.mjhmc_mem_info(self.handle, ctypes.byref(f), ctypes.byref(t)), self.lib)
        return int(f.value), int(t.value)

    def autocor(self, samples, linear=False):
        """Lag sums ``out[k] = sum_series sum_t x_t x_{t+k}`` of a host array [n_dims, n_batch, n_samples]
        (circular in time unless ``linear``); see mjhmc_autocor in include/mjhmc_hip.h."""
        samples = np.ascontiguousarray(samples, dtype=np.float64)
        assert samples.ndim == 3
        n = samples.shape[2]
        out = np.empty(n, dtype=np.float64)
        check(self.lib.mjhmc_autocor(self.handle, ptr(samples), int(samples.shape[0] * samples.shape[1]), int(n),
                                     1 if linear else 0, ptr(out)), self.lib)
        return out


    def draw_from(self, rates, unit_exp):
        """Waiting times ``(1 / rate) * e`` (inf where the rate is 0); returns (draws, first_bad) with first_bad = -1
        or the index of the first non-finite rate (mjhmc_draw_from in include/mjhmc_hip.h)."""
        rates = np.ascontiguousarray(rates, dtype=np.float64)
        unit_exp = np.ascontiguousarray(unit_exp, dtype=np.float64)
        assert rates.ndim == 1 and unit_exp.shape == rates.shape
        out = np.empty_like(rates)
        bad = ctypes.c_int64(-1)
        rc = self.lib.mjhmc_draw_from(self.handle, ptr(rates), ptr(unit_exp), int(rates.size), ptr(out), ctypes.byref(bad))
        if rc != _lib.ERR_NONFINITE:
            check(rc, self.lib)
        return out, int(bad.value)

    def min_idx(self, draws):
        """argmin over the rows of a (k, n) float64 array per column, first minimum on ties (mjhmc_min_idx)."""
        draws = np.ascontiguousarray(draws, dtype=np.float64)
        assert draws.ndim == 2 and draws.shape[0] >= 1
        which = np.empty(draws.shape[1], dtype=np.int32)
        check(self.lib.mjhmc_min_idx(self.handle, ptr(draws), int(draws.shape[0]), int(draws.shape[1]), ptr(which)), self.lib)
        return which


def context(device=0):
    """Process-wide context per device index."""
    if device not in _contexts:
        _contexts[device] = Context(device)
    return _contexts[device]


class DeviceEnergy(object):
    """(kind, ndims, float64 params) resident on one device."""

    def __init__(self, ctx, kind, ndims, params):
        self.ctx = ctx
        self.kind = int(kind)
        self.ndims = int(ndims)
        self.params = np.ascontiguousarray(np.atleast_1d(params), dtype=np.float64)
        h = ctypes.c_void_p()
        check(ctx.lib.mjhmc_energy_create(ctx.handle, self.kind, self.ndims, ptr(self.params), self.params.size,
                                          ctypes.byref(h)), ctx.lib)
        self.handle = h

    @classmethod
    def from_expr(cls, ctx, ndims, energy_expr, grad_expr, params=(), stats=(), energy0_expr=None):
        """An energy given as C expressions of ``x`` (coordinate), ``d`` (its index), ``p[k]`` (float64 parameters) and,
        when ``stats`` are given, ``S[k]`` (per-particle sums of the stat expressions):
            E(x) = energy0_expr(S) + sum_d energy_expr(x_d, d, S),   dE/dx_d = grad_expr(x_d, d, S);
        compiled with hipRTC around the engine's kernel templates (mjhmc_energy_create_expr[_coupled],
        include/mjhmc_hip.h)."""
        self = cls.__new__(cls)
        self.ctx = ctx
        self.kind = _lib.E_USER_EXPR
        self.ndims = int(ndims)
        self.params = np.ascontiguousarray(np.atleast_1d(np.asarray(params, dtype=np.float64)).ravel())
        stats = [stats] if isinstance(stats, str) else list(stats)
        h = ctypes.c_void_p()
        check(ctx.lib.mjhmc_energy_create_expr_coupled(
            ctx.handle, self.ndims, ';'.join(str(t) for t in stats).encode() if stats else None, str(energy_expr).encode(),
            str(energy0_expr).encode() if energy0_expr else None, str(grad_expr).encode(),
            ptr(self.params) if self.params.size else None, self.params.size, _lib.KERNEL_HEADERS.encode(), ctypes.byref(h)), ctx.lib)
        self.handle = h
        return self

    @classmethod
    def host(cls, ctx, ndims, energy_func, energy_grad_func):
        """An energy only the caller can evaluate (MJHMC_E_HOST, include/mjhmc_hip.h): two opaque Python callables
        ``energy_func(X (D,n)) -> (n,) or (1,n)`` and ``energy_grad_func(X) -> (D,n)`` (README.md:27-36).  Samplers of it
        (HostEnergySampler) keep the state and the whole jump process on the device and call back once per leapfrog step."""
        self = cls(ctx, _lib.E_HOST, ndims, np.zeros(0))
        self.energy_func, self.energy_grad_func = energy_func, energy_grad_func
        return self

    def host_E(self, X):
        e = np.asarray(self.energy_func(X), dtype=np.float64).reshape(-1)
        if e.shape != (X.shape[1],):
            raise ValueError('energy_func must return one energy per column: got %r for X %r' % (e.shape, X.shape))
        return np.ascontiguousarray(e)

    def host_grad(self, X):
        g = np.asarray(self.energy_grad_func(X), dtype=np.float64)
        if g.shape != X.shape:
            raise ValueError('energy_grad_func must return an array shaped like X: got %r for X %r' % (g.shape, X.shape))
        return np.ascontiguousarray(g)

    def eval(self, X, want_E=True, want_grad=True, dtype='float64'):
        X = as_f64(X)
        if X.ndim != 2 or X.shape[0] != self.ndims:
            raise ValueError('X must be (ndims, n)')
        if self.kind == _lib.E_HOST:                       # the callables ARE the energy
            return (self.host_E(X) if want_E else None), (self.host_grad(X) if want_grad else None)
        n = X.shape[1]
        E = np.empty(n) if want_E else None
        G = np.empty((self.ndims, n)) if want_grad else None
        if n:
            check(self.ctx.lib.mjhmc_eval(self.handle, dtype_code(dtype), ptr(X), n, ptr(E), ptr(G)), self.ctx.lib)
        return E, G

    def leapfrog(self, X, V, epsilon, n_steps, want_grad=True, dtype='float64'):
        """n_steps leapfrog steps from (X, V) in the reference's operation order (hmc_state.py:86-100).
        Returns (X', V', EX' (n,), EV' (n,), dEdX' or None)."""
        if self.kind == _lib.E_HOST:
            raise NotImplementedError('the stand-alone leapfrog operator needs a device energy; opaque callables have none')
        X = as_f64(X)
        V = as_f64(V, X.shape)
        if X.ndim != 2 or X.shape[0] != self.ndims:
            raise ValueError('X must be (ndims, n)')
        n = X.shape[1]
        Xo, Vo = np.empty_like(X), np.empty_like(X)
        EX, EV = np.empty(n), np.empty(n)
        G = np.empty_like(X) if want_grad else None
        if n:
            check(self.ctx.lib.mjhmc_leapfrog(self.handle, dtype_code(dtype), ptr(X), ptr(V), n, float(epsilon), int(n_steps),
                                              ptr(Xo), ptr(Vo), ptr(EX), ptr(EV), ptr(G)), self.ctx.lib)
        return Xo, Vo, EX, EV, G

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.ctx.lib.mjhmc_energy_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class DeviceSampler(object):
    """One shard of particle columns on one GPU."""

    def __init__(self, energy, Xinit, Vinit=None, seed=0, first_particle_id=0, dtype='float64',
                 mode=_lib.MODE_MJHMC):
        self.energy = energy
        self.ctx = energy.ctx
        self.lib = self.ctx.lib
        X = as_f64(Xinit)
        if X.ndim != 2 or X.shape[0] != energy.ndims:
            raise ValueError('Xinit must be (ndims, nparticles)')
        self.ndims, self.nparticles = X.shape
        V = None if Vinit is None else as_f64(Vinit, X.shape)
        h = ctypes.c_void_p()
        check(self.lib.mjhmc_sampler_create(self.ctx.handle, energy.handle, self.nparticles, int(first_particle_id),
                                            dtype_code(dtype), ptr(X), ptr(V), ctypes.c_uint64(int(seed) & (2 ** 64 - 1)),
                                            int(mode), ctypes.byref(h)), self.lib)
        self.handle = h
        self.ring_slots = 0

    def set_hparams(self, epsilon, num_leapfrog_steps, p_r, beta=1.0, p_flip=0.5):
        check(self.lib.mjhmc_set_hparams(self.handle, float(epsilon), int(num_leapfrog_steps), float(p_r), float(beta),
                                         float(p_flip)), self.lib)

    def iterate(self, n_iter=1, replay_normal=None, replay_exp=None, replay_unif=None, ring_slot0=-1):
        """Returns (list of IterStats for the attempts made, n_done)."""
        D, N = self.ndims, self.nparticles
        rn = None if replay_normal is None else as_f64(replay_normal).reshape(n_iter, D, N)
        re = None if replay_exp is None else as_f64(replay_exp).reshape(n_iter, 3, N)
        ru = None if replay_unif is None else as_f64(replay_unif).reshape(n_iter, 2 * N + 1)
        stats = (_lib.IterStats * n_iter)()
        done = ctypes.c_int()
        check(self.lib.mjhmc_iterate(self.handle, int(n_iter), ptr(rn), ptr(re), ptr(ru), int(ring_slot0), stats,
                                     ctypes.byref(done)), self.lib)
        n_done = done.value
        attempts = n_done + 1 if n_done < n_iter else n_iter
        return [stats[i] for i in range(attempts)], n_done

    def iterate_download(self, n_iter, ring_slot0, out, k0=0):
        """n_iter iterations into ring slots [ring_slot0, ring_slot0 + n_iter), each slot brought to
        out[:, (k0 + i) * N : (k0 + i + 1) * N] while the following iterations run (mjhmc_iterate_download).  ``out``: the
        C-contiguous float64 array (ndims, n_total * nparticles) of the whole run.  Returns (stats, n_done)."""
        D, N = self.ndims, self.nparticles
        if out.dtype != np.float64 or not out.flags.c_contiguous or out.ndim != 2 or out.shape[0] != D or out.shape[1] % N:
            raise ValueError('out must be a C-contiguous float64 array (ndims, n_total * nparticles)')
        stats = (_lib.IterStats * n_iter)()
        done = ctypes.c_int()
        check(self.lib.mjhmc_iterate_download(self.handle, int(n_iter), int(ring_slot0), ptr(out), out.shape[1] // N, int(k0), stats,
                                              ctypes.byref(done)), self.lib)
        n_done = done.value
        attempts = n_done + 1 if n_done < n_iter else n_iter
        return [stats[i] for i in range(attempts)], n_done

    def ring_budget_slots(self, n_wanted, share=0.6, staging=True):
        """How many whole-state ring slots (of n_wanted) the device can take: those it already has, or `share` of the free
        memory -- at least 2 (an iteration reads one slot and writes the next).  ``staging``: every slot also gets a staging
        copy in the host layout (mjhmc_iterate_download: the streamed sample()); a ring that is recorded and then read or
        gathered from needs none."""
        b = ctypes.c_uint64()
        check(self.lib.mjhmc_ring_slot_bytes(self.handle, ctypes.byref(b)), self.lib)
        per = int(b.value) + (8 * self.ndims * self.nparticles if staging else 0)
        free, _ = self.ctx.mem_info()
        fit = max(int(share * free // per) + self.ring_slots, 2)
        return min(int(n_wanted), fit)

    def reset_flf_cache(self):
        check(self.lib.mjhmc_reset_flf_cache(self.handle), self.lib)

    def checkpoint(self):
        check(self.lib.mjhmc_checkpoint(self.handle), self.lib)

    def restore(self):
        check(self.lib.mjhmc_restore(self.handle), self.lib)

    def rollback(self):
        check(self.lib.mjhmc_rollback(self.handle), self.lib)

    def get_tick(self):
        t = ctypes.c_uint64()
        check(self.lib.mjhmc_get_tick(self.handle, ctypes.byref(t)), self.lib)
        return int(t.value)

    def set_tick(self, tick):
        check(self.lib.mjhmc_set_tick(self.handle, ctypes.c_uint64(int(tick))), self.lib)

    def advance_tick(self, n=1):
        check(self.lib.mjhmc_advance_tick(self.handle, int(n)), self.lib)

    def read(self, field, out=None):
        D, N = self.ndims, self.nparticles
        if out is not None:
            if field not in (_lib.F_X, _lib.F_V, _lib.F_DEDX) or out.shape != (D, N) or out.dtype != np.float64 or not out.flags.c_contiguous:
                raise ValueError('out: a C-contiguous float64 (ndims, nparticles) array for a matrix field')
        elif field in (_lib.F_X, _lib.F_V, _lib.F_DEDX):
            out = np.empty((D, N))
        elif field in (_lib.F_CACHE, _lib.F_TRANS):
            out = np.empty(N, dtype=np.uint8)
        else:
            out = np.empty(N)
        check(self.lib.mjhmc_read(self.handle, int(field), ptr(out), out.nbytes), self.lib)
        return out

    def write(self, field, arr):
        if field == _lib.F_HFLF:
            a = as_f64(np.asarray(arr, dtype=np.float64).reshape(-1), (self.nparticles,))
        else:
            a = as_f64(arr, (self.ndims, self.nparticles))
        check(self.lib.mjhmc_write(self.handle, int(field), ptr(a), a.nbytes), self.lib)

    def ring_alloc(self, n_slots):
        check(self.lib.mjhmc_ring_alloc(self.handle, int(n_slots)), self.lib)
        self.ring_slots = max(self.ring_slots, int(n_slots))

    def ring_read_dwell(self, slot0, n):
        out = np.empty((n, self.nparticles))
        check(self.lib.mjhmc_ring_read_dwell(self.handle, int(slot0), int(n), ptr(out)), self.lib)
        return out

    def ring_gather(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        out = np.empty((self.ndims, idx.size))
        if idx.size:
            check(self.lib.mjhmc_ring_gather(self.handle, ptr(idx), idx.size, ptr(out)), self.lib)
        return out

    def ring_read(self, slot0, n, stacked=False, out=None):
        """``out``: a caller-owned C-contiguous float64 array of the result's shape to fill (a ring of samples is GBs: a
        fresh array per call costs an allocation and a page fault per 4 KiB of it)."""
        shape = (self.ndims, self.nparticles, n) if stacked else (self.ndims, n * self.nparticles)
        if out is None:
            out = np.empty(shape)
        elif out.shape != shape or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError('out must be a C-contiguous float64 array of shape %r' % (shape,))
        check(self.lib.mjhmc_ring_read(self.handle, int(slot0), int(n), 1 if stacked else 0, ptr(out)), self.lib)
        return out

    def ring_moments(self, slot0, n, shift=0.0):
        """(sum (x - shift), sum (x - shift)^2) over all state elements of ring slots [slot0, slot0 + n)."""
        a, b = ctypes.c_double(), ctypes.c_double()
        check(self.lib.mjhmc_ring_moments(self.handle, int(slot0), int(n), float(shift), ctypes.byref(a), ctypes.byref(b)), self.lib)
        return a.value, b.value

    def ring_autocor(self, slot0, n, linear=False):
        """Lag sums over time of ring slots [slot0, slot0 + n), summed over all state elements."""
        out = np.empty(int(n), dtype=np.float64)
        check(self.lib.mjhmc_ring_autocor(self.handle, int(slot0), int(n), 1 if linear else 0, ptr(out)), self.lib)
        return out

    def last_timing(self):
        t, k, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        check(self.lib.mjhmc_last_timing(self.handle, ctypes.byref(t), ctypes.byref(k), ctypes.byref(n)), self.lib)
        return dict(total_ms=t.value, jump_kernel_ms=k.value, n_jump_launches=n.value)

    def set_timing(self, on):
        """the HIP-event pair around a call's launches (last_timing): ~8 us of every call; on by default"""
        check(self.lib.mjhmc_set_timing(self.handle, 1 if on else 0), self.lib)

    def sync(self):
        check(self.lib.mjhmc_sync(self.handle), self.lib)

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.mjhmc_sampler_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostEnergySampler(DeviceSampler):
    """DeviceSampler for an energy given as opaque Python callables (DeviceEnergy.host): ``iterate`` has the contract of
    the device's mjhmc_iterate, but every sampling iteration is driven from here through mjhmc_traj_begin / _step /
    _finish -- the callables are called once per leapfrog step on the proposal columns (N + n_cold of them), everything
    else (state, leapfrog arithmetic, jump decision, commit, counters, sample ring) runs on the device."""

    def __init__(self, energy, Xinit, Vinit=None, seed=0, first_particle_id=0, dtype='float64', mode=_lib.MODE_MJHMC):
        if dtype_code(dtype) != _lib.F64:
            raise ValueError('host-evaluated energies run in float64')
        super(HostEnergySampler, self).__init__(energy, Xinit, Vinit, seed, first_particle_id, 'float64', mode)
        self._L = 5
        self._set_energy(as_f64(Xinit))

    def _set_energy(self, X):
        E, G = self.energy.host_E(X), self.energy.host_grad(X)
        check(self.lib.mjhmc_host_set_energy(self.handle, ptr(E), ptr(G)), self.lib)

    def set_hparams(self, epsilon, num_leapfrog_steps, p_r, beta=1.0, p_flip=0.5):
        super(HostEnergySampler, self).set_hparams(epsilon, num_leapfrog_steps, p_r, beta, p_flip)
        self._L = int(num_leapfrog_steps)

    def write(self, field, arr):
        super(HostEnergySampler, self).write(field, arr)
        if field == _lib.F_X:                              # HMCState(X): E and dE/dX of the new positions (hmc_state.py:30-38)
            self._set_energy(as_f64(arr, (self.ndims, self.nparticles)))

    def _attempt(self, rn, re, ru, ring_slot):
        n = ctypes.c_int64()
        check(self.lib.mjhmc_traj_begin(self.handle, ctypes.byref(n)), self.lib)
        X = np.empty((self.ndims, n.value))
        g = None
        for _ in range(self._L):
            check(self.lib.mjhmc_traj_step(self.handle, ptr(g), 0, ptr(X)), self.lib)
            g = self.energy.host_grad(X)
        check(self.lib.mjhmc_traj_step(self.handle, ptr(g), 1, None), self.lib)
        if self._L == 0:                                   # no step was taken: the end points are the start points
            X0 = self.read(_lib.F_X)
            X = np.concatenate([X0, X0[:, np.isnan(self.read(_lib.F_HFLF))]], axis=1)[:, :n.value]
        E = self.energy.host_E(X)
        st = _lib.IterStats()
        check(self.lib.mjhmc_traj_finish(self.handle, ptr(E), ptr(rn), ptr(re), ptr(ru), int(ring_slot), ctypes.byref(st)),
              self.lib)
        return st

    def iterate(self, n_iter=1, replay_normal=None, replay_exp=None, replay_unif=None, ring_slot0=-1):
        D, N = self.ndims, self.nparticles
        rn = None if replay_normal is None else as_f64(replay_normal).reshape(n_iter, D, N)
        re = None if replay_exp is None else as_f64(replay_exp).reshape(n_iter, 3, N)
        ru = None if replay_unif is None else as_f64(replay_unif).reshape(n_iter, 2 * N + 1)
        stats, done = [], 0
        for i in range(n_iter):
            st = self._attempt(None if rn is None else np.ascontiguousarray(rn[i]), None if re is None else np.ascontiguousarray(re[i]),
                               None if ru is None else np.ascontiguousarray(ru[i]), ring_slot0 + i if ring_slot0 >= 0 else -1)
            stats.append(st)
            if st.nonfinite:
                break
            done += 1
        return stats, done

    def iterate_download(self, n_iter, ring_slot0, out, k0=0):
        """(the caller's callables pace this sampler: the slots are read once the iterations are done)"""
        stats, done = self.iterate(n_iter, ring_slot0=ring_slot0)
        N = self.nparticles
        for i in range(done):
            out[:, (k0 + i) * N:(k0 + i + 1) * N] = self.ring_read(ring_slot0 + i, 1)
        return stats, done

    def last_timing(self):
        return dict(total_ms=0.0, jump_kernel_ms=0.0, n_jump_launches=0)


def make_sampler(energy, *args, **kwargs):
    """DeviceSampler, or HostEnergySampler for an energy only the caller can evaluate."""
    cls = HostEnergySampler if energy.kind == _lib.E_HOST else DeviceSampler
    return cls(energy, *args, **kwargs)
