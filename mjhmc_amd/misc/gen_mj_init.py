"""Fair-initialisation generator (mirrors mjhmc/misc/gen_mj_init.py:14-98).

The continuous-time samplers run the *embedded* Markov chain, whose stationary distribution differs
from the target's, so the reference burns a sampler in for 1e6 steps at 1000 particles and caches the
end points (``initializations/{Class}_{hash}.pickle`` = ``(mjhmc_endpt, emc_var, true_var,
control_endpt)``).  This is the heaviest consumer of the hot path; here the burn-in is batched device
work and the online variance is reduced on the device (``mjhmc_ring_moments``), slot block by slot block.

Differences kept deliberate: the cache key is a stable SHA-1 of the class name and the device-energy
parameters (Python-2 tuple hashes are not reproducible), and the step counts are arguments.
"""
import hashlib
import os
import pickle

import numpy as np

from ..samplers.markov_jump_hmc import MarkovJumpHMC, ControlHMC

BURN_IN_STEPS = int(1E6)
VAR_STEPS = int(5E5)
MAX_N_PARTICLES = 1000


def online_variance(sampler, distribution, var_steps=VAR_STEPS, block=256):
    """Unbiased variance of every value of ``var_steps`` consecutive ``sample(1)`` states
    (gen_mj_init.py:76-98: Welford over ``sampler.sample(1).ravel()``), accumulated on the device in
    blocks of ``block`` steps and merged with the pairwise update of Chan et al.
    Returns (variance estimate, sampler)."""
    count, mean, m2 = 0, 0.0, 0.0
    per_step = distribution.nbatch * distribution.ndims
    done = 0
    sampler._dev.ring_alloc(block)
    while done < var_steps:
        k = min(block, var_steps - done)
        sampler._run(k, ring_slot0=0)
        s1, s2 = sampler._dev.ring_moments(0, k, shift=mean)      # sums of (x - mean_so_far)
        if sampler._comm is not None:
            raise NotImplementedError('online_variance runs unsharded (the generator uses <= 1000 particles)')
        nb = k * per_step
        mean_b = mean + s1 / nb
        m2_b = s2 - s1 * s1 / nb
        delta = mean_b - mean
        tot = count + nb
        m2 = m2 + m2_b + delta * delta * count * nb / tot
        mean = mean + delta * nb / tot
        count = tot
        done += k
    sampler._publish()
    if hasattr(sampler, '_read_dwell'):
        sampler._read_dwell()
    return m2 / float(var_steps * distribution.nbatch * distribution.ndims - 1), sampler


def generate_initialization(distribution, burn_in_steps=BURN_IN_STEPS, var_steps=VAR_STEPS, seed=None):
    """gen_mj_init.py:14-52.  Returns (mjhmc_endpt, emc_var_estimate, true_var_estimate, control_endpt)."""
    assert burn_in_steps > var_steps
    distribution.generation_instance = True
    mjhmc = MarkovJumpHMC(distribution=distribution, resample=False, seed=seed)
    mjhmc._run(burn_in_steps - var_steps)
    mjhmc._publish()
    emc_var_estimate, mjhmc = online_variance(mjhmc, distribution, var_steps)
    mjhmc_endpt = mjhmc.state.copy().X                      # V is discarded: p(x, v) = p(x) p(v)

    distribution.mjhmc = False
    try:
        distribution.gen_init_X()
    except NotImplementedError:
        print("No explicit init method found, using mjhmc endpoint")
        distribution.Xinit = mjhmc_endpt
    distribution.E_count = 0
    distribution.dEdX_count = 0
    distribution.generation_instance = False
    # ControlHMC resets the distribution (init_X); generate from the state set above
    keep = distribution.Xinit
    distribution.init_X = lambda: setattr(distribution, 'Xinit', keep)
    try:
        control = ControlHMC(distribution=distribution, seed=None if seed is None else seed + 1)
    finally:
        del distribution.init_X
    control._run(burn_in_steps - var_steps)
    control._publish()
    true_var_estimate, control = online_variance(control, distribution, var_steps)
    control_endpt = control.state.copy().X
    return mjhmc_endpt, emc_var_estimate, true_var_estimate, control_endpt


def stable_digest(distribution):
    kind, params = distribution.device_energy()
    h = hashlib.sha1()
    h.update(type(distribution).__name__.encode())
    h.update(np.int64([kind, distribution.ndims]).tobytes())
    if isinstance(params, tuple) and params and isinstance(params[0], str):     # user expressions + their parameters
        for part in params:
            if part is None:
                continue
            if isinstance(part, (list, tuple)):
                part = ';'.join(str(t) for t in part)
            h.update(part.encode() if isinstance(part, str) else np.ascontiguousarray(part, dtype=np.float64).tobytes())
    elif isinstance(params, (tuple, list)) and any(callable(q) for q in params):
        # opaque callables (LambdaDistribution on MJHMC_E_HOST): no bytes to hash -- the name the distribution was given
        # stands for them, as in Distribution.__hash__ of the reference (distributions.py:247-251)
        h.update(str(getattr(distribution, 'name', '')).encode())
    else:
        h.update(np.ascontiguousarray(params, dtype=np.float64).tobytes())
    # the arithmetic the burn-in ran in, where the distribution offers a choice (ProductOfT: float64 / float32 state,
    # SparseImageCode: float32 / bfloat16): its end points are those of THAT chain
    dt = getattr(distribution, 'state_dtype', 'float64')
    if dt != 'float64':
        h.update(str(dt).encode())
    return h.hexdigest()[:16]


def cache_initialization(distribution, directory, **kwargs):
    """gen_mj_init.py:54-73 with a reproducible file name.  Returns the path written."""
    result = generate_initialization(distribution, **kwargs)
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, '{}_{}.pickle'.format(type(distribution).__name__, stable_digest(distribution)))
    with open(path, 'wb') as cache_file:
        pickle.dump(result, cache_file, protocol=2)     # what _ArraysOnlyUnpickler (and the reference's Python 2) reads
    return path


class _ArraysOnlyUnpickler(pickle.Unpickler):
    """Unpickling runs whatever callables the file names.  The cache files hold tuples of NumPy arrays and floats:
    only the three globals an ndarray pickle needs are resolved, anything else is refused."""

    _ALLOWED = {('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
                ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'),
                ('numpy', 'ndarray'), ('numpy', 'dtype'),
                ('numpy.core.numeric', '_frombuffer'), ('numpy._core.numeric', '_frombuffer'),   # protocol-5 array pickles
                ('_codecs', 'encode')}      # how protocol-2 pickles written by Python 3 carry an array's bytes

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super(_ArraysOnlyUnpickler, self).find_class(module, name)
        raise pickle.UnpicklingError('initialisation cache: refusing to resolve %s.%s (arrays and floats only)' % (module, name))


def load_reference_initialization(path):
    """A cache file written by the REFERENCE (mjhmc/misc/gen_mj_init.py:54-73 under Python 2, e.g. the files it ships in
    initializations/): returns (mjhmc_endpt, emc_var_estimate, true_var_estimate, control_endpt); older files hold only
    the first three (control_endpt is then None).  The file is read with an unpickler that resolves nothing but
    ndarray / dtype reconstruction, so a crafted file cannot run code; still, load files you trust."""
    with open(path, 'rb') as cache_file:
        t = _ArraysOnlyUnpickler(cache_file, encoding='latin1').load()   # Python 2 str -> bytes of the ndarray payloads
    if len(t) == 4:
        mj, emc_var, true_var, ctl = t
    elif len(t) == 3:
        (mj, emc_var, true_var), ctl = t, None
    else:
        raise ValueError('%s: expected a 3- or 4-tuple, got %d entries' % (path, len(t)))
    return (np.asarray(mj, dtype=np.float64), float(emc_var), float(true_var),
            None if ctl is None else np.asarray(ctl, dtype=np.float64))


def load_initialization(distribution, directory):
    """Distribution.load_cache (mjhmc/misc/distributions.py:182-195)."""
    path = os.path.join(directory, '{}_{}.pickle'.format(type(distribution).__name__, stable_digest(distribution)))
    with open(path, 'rb') as cache_file:
        return _ArraysOnlyUnpickler(cache_file).load()
