"""Energy-model interface of the drop-in (mirrors mjhmc/misc/distributions.py and the energy
formulas of mjhmc/misc/tf_distributions.py).

A ``Distribution`` here is a *description*: ``device_energy()`` names the HIP functor and its
parameters; evaluation (``E`` / ``dEdX``) and sampling run on the MI355X.  What is kept from the
reference: the constructor signatures, ``E/dEdX`` counting facades (distributions.py:62-75),
``Xinit/ndims/nbatch/mjhmc/generation_instance/backend/max_n_particles``, ``reset()``,
``__call__`` (NUTS convention, :172-180) and the per-class ``gen_init_X`` recipes.

Out of scope: the 1e6-step "fair initialisation" pickle cache behind
the reference's ``init_X`` (:96-149).  ``init_X`` here uses ``gen_init_X`` directly; pass your
own fair ``Xinit`` through ``LambdaDistribution(init=...)`` or by overriding ``init_X``.
"""
import numpy as np

from .. import _lib
from .. import engine


class Distribution(object):
    """Interface class; subclass it and provide ``device_energy`` (distributions.py:13-59)."""

    def __init__(self, ndims=2, nbatch=100):
        self.ndims = ndims
        self.nbatch = nbatch
        if not hasattr(self, 'backend'):
            self.backend = 'hip'
        self.mjhmc = None
        self.E_count = 0
        self.dEdX_count = 0
        self.generation_instance = False
        if not hasattr(self, 'state_dtype'):
            self.state_dtype = 'float64'         # arithmetic type of state and force on the device
        if not hasattr(self, 'max_n_particles'):
            self.max_n_particles = None
        self._dev = None
        self.init_X()

    # -- device binding -------------------------------------------------------------------
    def device_energy(self):
        """Return (kind, float64 parameter vector) of the HIP functor for this energy."""
        raise NotImplementedError('%s has no device functor: implement device_energy()' % type(self).__name__)

    def bind(self, device=0):
        """DeviceEnergy for this distribution on ``device`` (cached)."""
        if self._dev is None or self._dev.ctx.device != device:
            kind, params = self.device_energy()
            if kind == _lib.E_USER_EXPR:                   # params = (energy_expr, grad_expr, float64 parameters, stats, energy0_expr)
                self._dev = engine.DeviceEnergy.from_expr(engine.context(device), self.ndims, *params)
            elif kind == _lib.E_HOST:                      # params = (energy_func, energy_grad_func): opaque callables
                self._dev = engine.DeviceEnergy.host(engine.context(device), self.ndims, *params)
            else:
                self._dev = engine.DeviceEnergy(engine.context(device), kind, self.ndims, params)
        return self._dev

    # -- reference API --------------------------------------------------------------------
    def E(self, X):
        self.E_count += X.shape[1]
        return self.E_val(X)

    def E_val(self, X):
        E, _ = self.bind().eval(X, want_E=True, want_grad=False, dtype=self.state_dtype)
        return E.reshape((1, -1))

    def dEdX(self, X):
        self.dEdX_count += X.shape[1]
        return self.dEdX_val(X)

    def dEdX_val(self, X):
        _, G = self.bind().eval(X, want_E=False, want_grad=True, dtype=self.state_dtype)
        return G

    def __hash__(self):
        raise NotImplementedError()

    def init_X(self):
        try:
            self.gen_init_X()
        except NotImplementedError:
            self.Xinit = np.random.randn(self.ndims, self.nbatch)      # distributions.py:137-139

    def gen_init_X(self):
        raise NotImplementedError()

    # -- fair-initialisation cache (distributions.py:104-149, 182-195) ----------------------------
    @staticmethod
    def _cache_directory(directory=None):
        import os
        return directory or os.environ.get('MJHMC_INIT_CACHE') or os.path.join(
            os.path.expanduser('~'), '.cache', 'mjhmc_amd', 'initializations')

    def cached_init_X(self, directory=None, **generator_kwargs):
        """Set ``Xinit`` from the cached burn-in end points of this distribution (MJHMC end points when the
        distribution serves a continuous-time sampler, ControlHMC's otherwise), generating and caching them first
        when the cache file does not exist -- the reference's ``init_X`` (distributions.py:96-149).  Unlike the
        reference this is opt-in: ``init_X`` here never starts a 10^6-step burn-in behind the caller's back.
        ``generator_kwargs`` (burn_in_steps, var_steps, seed) go to gen_mj_init.generate_initialization."""
        import os
        from . import gen_mj_init as G
        directory = self._cache_directory(directory)
        name = '{}_{}.pickle'.format(type(self).__name__, G.stable_digest(self))
        if not os.path.exists(os.path.join(directory, name)):
            old_nbatch, old_mjhmc = self.nbatch, self.mjhmc
            self.nbatch = self.max_n_particles or G.MAX_N_PARTICLES
            self.generation_instance = True
            try:
                self.gen_init_X()                                     # start from the biased initialisation
            except NotImplementedError:
                self.Xinit = np.random.randn(self.ndims, self.nbatch)    # "completely arbitrary choice" (:137-139)
            try:
                G.cache_initialization(self, directory, **generator_kwargs)
            finally:
                self.nbatch, self.mjhmc = old_nbatch, old_mjhmc
                self.generation_instance = False
        mjhmc_endpt, _, _, control_endpt = G.load_initialization(self, directory)
        self.Xinit = (mjhmc_endpt if self.mjhmc else control_endpt)[:, :self.nbatch]

    def load_cache(self, directory=None):
        """(mjhmc_endpt, emc_var_estimate, true_var_estimate, control_endpt) of this distribution's cache file;
        raises if it does not exist (distributions.py:182-195)."""
        from . import gen_mj_init as G
        return G.load_initialization(self, self._cache_directory(directory))

    def reset(self):
        self.E_count = 0
        self.dEdX_count = 0
        if not self.generation_instance:
            self.init_X()
        return self

    def __call__(self, X):
        rshp_X = X.reshape(len(X), 1)
        E = float(np.asarray(self.E(rshp_X)).reshape(()))
        dEdX = self.dEdX(rshp_X).T[0]
        return -E, -dEdX


class LambdaDistribution(Distribution):
    """'Anonymous' distribution (distributions.py:198-251, README.md:27-36).

    The reference's class ignores its callables (it evaluates a non-existent ``self.J``); the README contract is
    what is kept: ``energy_func`` / ``energy_grad_func`` define the distribution and ``init`` is the initial state.
    Python callables cannot run inside a GPU kernel, so the callables reach the device in one of four ways:

    * nothing, and the callables are recognised: they are probed on a few points and matched against the built-in
      elementwise families (isotropic Gaussian with any sigma -- the README example -- and diagonal Gaussians);
    * nothing, and they are NOT recognised (any opaque pair of callables, e.g. a dense quadratic form): the sampler
      keeps the state and the whole jump process on the device and calls ``energy_grad_func`` back once per leapfrog
      step and ``energy_func`` once per iteration (MJHMC_E_HOST, mjhmc_traj_* of include/mjhmc_hip.h) -- correct for
      any callables, slow by construction (a host round trip per leapfrog step);
    * ``device_expr=(energy_expr, grad_expr)`` [+ ``device_params``]: C expressions of one coordinate for a separable
      energy ``E(x) = sum_d energy_expr(x_d)``, ``dE/dx_d = grad_expr(x_d)`` with ``x`` the coordinate, ``d`` its
      index and ``p[k]`` the float64 ``device_params``; the engine's kernels are compiled around them with hipRTC
      (mjhmc_energy_create_expr).  When callables are given as well they are checked against the compiled energy.
      Coupled coordinates: ``device_expr=dict(stats=[...], energy=..., energy0=..., grad=...)`` -- the ``stats``
      expressions are summed over a particle's coordinates into ``S[k]``, which the other expressions may use:
      ``E = energy0(S) + sum_d energy(x_d, d, S)``, ``dE/dx_d = grad(x_d, d, S)`` (mjhmc_energy_create_expr_coupled);
    * ``device_energy=(kind, params)``: one of the built-in device energies by name.
    """

    def __init__(self, energy_func=None, energy_grad_func=None, init=None, name=None, device_energy=None,
                 device_expr=None, device_params=()):
        self.energy_func = energy_func
        self.energy_grad_func = energy_grad_func
        self.init = np.array(init, dtype=np.float64)
        self.name = name or str(np.random.random())
        self._functor = device_energy
        self._checked = False
        if device_expr is not None:
            if isinstance(device_expr, dict):              # coupled through per-particle statistics S[k]
                e_expr, g_expr = device_expr['energy'], device_expr['grad']
                stats = device_expr.get('stats', ())
                stats = [stats] if isinstance(stats, str) else [str(t) for t in stats]
                e0 = device_expr.get('energy0')
            else:
                (e_expr, g_expr), stats, e0 = device_expr, [], None
            self._functor = (_lib.E_USER_EXPR, (str(e_expr), str(g_expr), np.asarray(device_params, dtype=np.float64),
                                                stats, None if e0 is None else str(e0)))
        elif self._functor is None:
            self._functor = _recognise(energy_func, energy_grad_func, self.init.shape[0])
            if self._functor is None and energy_func is not None and energy_grad_func is not None:
                self._functor = (_lib.E_HOST, (energy_func, energy_grad_func))       # opaque callables: host call-backs
            self._checked = True
        super(LambdaDistribution, self).__init__(ndims=self.init.shape[0], nbatch=self.init.shape[1])

    def device_energy(self):
        if self._functor is None:
            raise NotImplementedError('LambdaDistribution %r needs energy_func and energy_grad_func (or device_expr= / '
                                      'device_energy=)' % self.name)
        return self._functor

    def bind(self, device=0):
        dev = super(LambdaDistribution, self).bind(device)
        if not self._checked and self.energy_func is not None and self.energy_grad_func is not None:
            # the callables are the contract (README.md:27-36): the device energy must be the same function
            self._checked = True
            P = np.random.RandomState(12345).randn(self.ndims, 7)
            E, G = dev.eval(P)
            e = np.asarray(self.energy_func(P), dtype=np.float64).reshape(-1)
            g = np.asarray(self.energy_grad_func(P), dtype=np.float64)
            if not (np.allclose(E, e, rtol=1e-9, atol=1e-12) and np.allclose(G, g, rtol=1e-9, atol=1e-12)):
                raise ValueError('LambdaDistribution %r: the device energy disagrees with energy_func / '
                                 'energy_grad_func (max |dE| %g, max |d grad| %g)'
                                 % (self.name, np.abs(E - e).max(), np.abs(G - g).max()))
        return dev

    def gen_init_X(self):
        self.Xinit = self.init

    def __hash__(self):
        return hash((self.ndims, self.nbatch, self.name))


def _recognise(energy_func, grad_func, ndims):
    """Match user callables against the built-in Gaussian families by probing (no sampling is ever done with the
    callables themselves): g(x) = j * x elementwise with E = sum j x^2 / 2 -- isotropic when all j are equal."""
    if energy_func is None or grad_func is None:
        return None
    rng = np.random.RandomState(12345)
    P = rng.randn(ndims, 5)
    try:
        g = np.asarray(grad_func(P), dtype=np.float64)
        e = np.asarray(energy_func(P), dtype=np.float64).reshape(-1)
    except Exception:
        return None
    if g.shape != P.shape or e.shape != (5,):
        return None
    ratio = g / P
    j = np.median(ratio, axis=1)
    if np.any(j <= 0) or not np.allclose(ratio, j[:, None], rtol=1e-12, atol=0):
        return None
    if not np.allclose(e, 0.5 * np.sum(j[:, None] * P ** 2, axis=0), rtol=1e-12, atol=0):
        return None
    if np.allclose(j, j[0], rtol=1e-12, atol=0):
        return (_lib.E_ISO_GAUSS, np.array([1.0 / np.sqrt(float(np.median(j)))]))
    return (_lib.E_DIAG_GAUSS, j)


class Gaussian(Distribution):
    """Ill-conditioned Gaussian of the LAHMC paper (distributions.py:256-281)."""

    def __init__(self, ndims=2, nbatch=100, log_conditioning=6):
        self.conditioning = 10 ** np.linspace(-log_conditioning, 0, ndims)
        self.J = np.diag(self.conditioning)
        self.description = '%dD Anisotropic Gaussian, %g self.conditioning' % (ndims, 10 ** log_conditioning)
        super(Gaussian, self).__init__(ndims, nbatch)

    def device_energy(self):
        return (_lib.E_DIAG_GAUSS, np.asarray(self.conditioning, dtype=np.float64))

    def gen_init_X(self):
        self.Xinit = (1. / np.sqrt(self.conditioning).reshape((-1, 1))) * np.random.randn(self.ndims, self.nbatch)

    def __hash__(self):
        return hash((self.ndims, hash(tuple(self.conditioning))))


class RoughWell(Distribution):
    """distributions.py:283-312."""

    def __init__(self, ndims=2, nbatch=100, scale1=100, scale2=4):
        self.scale1 = scale1
        self.scale2 = scale2
        self.description = '{} Rough Well'.format(ndims)
        super(RoughWell, self).__init__(ndims, nbatch)

    def device_energy(self):
        return (_lib.E_ROUGH_WELL, np.array([self.scale1, self.scale2], dtype=np.float64))

    def gen_init_X(self):
        self.Xinit = self.scale1 * np.random.randn(self.ndims, self.nbatch)

    def __hash__(self):
        return hash((self.ndims, self.scale1, self.scale2))


class MultimodalGaussian(Distribution):
    """distributions.py:314-346 (as coded: modes at -/+ 2*separation along the first axis)."""

    def __init__(self, ndims=2, nbatch=100, separation=3):
        self.separation = separation
        self.sep_vec = np.array([separation] * nbatch + [0] * (ndims - 1) * nbatch).reshape(ndims, nbatch)
        self.sep_vec[0] += separation
        super(MultimodalGaussian, self).__init__(ndims, nbatch)

    def device_energy(self):
        return (_lib.E_MM_GAUSS, np.array([self.separation], dtype=np.float64))

    def init_X(self):
        self.Xinit = ((np.random.randn(self.ndims, self.nbatch) + self.sep_vec) +
                      (np.random.randn(self.ndims, self.nbatch) - self.sep_vec))

    def __hash__(self):
        return hash((self.ndims, self.separation))


class TestGaussian(Distribution):
    """Unit-variance (or sigma) isotropic Gaussian (distributions.py:348-370)."""
    __test__ = False  # not a pytest class

    def __init__(self, ndims=2, nbatch=100, sigma=1.):
        self.sigma = sigma
        super(TestGaussian, self).__init__(ndims, nbatch)

    def device_energy(self):
        return (_lib.E_ISO_GAUSS, np.array([self.sigma], dtype=np.float64))

    def gen_init_X(self):
        self.Xinit = np.random.randn(self.ndims, self.nbatch)

    def __hash__(self):
        return hash((self.ndims, self.sigma))


class TFGaussian(TestGaussian):
    """The reference's TensorFlow twin of TestGaussian (mjhmc/misc/tf_distributions.py:179-201; float32 there because the
    placeholder is, `state_dtype='float32'` selects the same here); extra keyword arguments (``device``, ...) are the
    TensorFlow plumbing of the reference and are accepted and ignored."""

    def __init__(self, ndims=2, nbatch=100, sigma=1., state_dtype='float64', **kwargs):
        self.state_dtype = state_dtype
        super(TFGaussian, self).__init__(ndims=ndims, nbatch=nbatch, sigma=sigma)


class Funnel(Distribution):
    """Neal's funnel (mjhmc/misc/tf_distributions.py:142-177).

    ``literal=False`` (default): the density the reference documents, x0 ~ N(0, scale^2),
    x_k ~ N(0, e^{x0}).  ``literal=True``: the energy exactly as the reference codes it (:161-165),
    which is the negated un-normalised log-density -- chains diverge on it, kept for one-shot
    E/dEdX parity only.
    """

    def __init__(self, scale=1.0, nbatch=50, ndims=10, literal=False):
        self.scale = float(scale)
        self.literal = literal
        super(Funnel, self).__init__(ndims, nbatch)

    def device_energy(self):
        return (_lib.E_FUNNEL_REF if self.literal else _lib.E_FUNNEL_NEAL, np.array([self.scale]))

    def gen_init_X(self):
        x_0 = np.random.normal(scale=self.scale, size=(1, self.nbatch))
        # exact draw from the funnel (the reference passes exp(x_0) as the std, :171-173)
        x_k = np.random.normal(scale=np.exp(x_0 / 2.), size=(self.ndims - 1, self.nbatch))
        self.Xinit = np.vstack((x_0, x_k))

    def __hash__(self):
        return hash((self.scale, self.ndims))


class ProductOfT(Distribution):
    """Product of Student-t experts (distributions.py:373-453).  The reference builds E and its
    gradient with Theano in float32; here both GEMMs of the gradient run on the MI355X matrix
    cores (exact-f32 MFMA).  ndims == nbasis, as the reference's initialiser requires (:391-392); any size (up to 512 dims
    on the register-resident tile kernels, beyond that block by block on the engine's multi-pass path).

    ``state_dtype``: 'float64' (default) is the reference's own arithmetic -- float64 ``HMCState`` arrays around the
    float32 force (:408-415 with hmc_state.py:29-38) -- on the tile kernel with the state streamed through its epilogue
    (csrc/dense_pot64.hip); 'float32' (extension, BASELINE configs[2]'s wording) keeps the particle state in float32
    too: the fused float32 tile kernel, ~7 % faster."""

    def __init__(self, ndims=36, nbasis=36, nbatch=100, lognu=None, W=None, b=None, state_dtype='float64'):
        if ndims != nbasis:
            raise NotImplementedError("Initializer only works for ndims == nbasis")
        if state_dtype not in ('float32', 'float64'):
            raise ValueError("ProductOfT state_dtype must be 'float32' or 'float64'")
        self.nbasis = nbasis
        self.state_dtype = state_dtype
        self.backend = 'hip-mfma'
        if W is None:
            W = np.eye(ndims, nbasis)
        self.weights = np.array(W, dtype='float32')
        pre_nu = np.random.rand(nbasis,) * 2 + 2.1 if lognu is None else np.exp(lognu)
        self.nu = np.array(pre_nu, dtype='float32')
        self.bias = np.array(np.zeros((nbasis,)) if b is None else b, dtype='float32')
        super(ProductOfT, self).__init__(ndims, nbatch)

    def device_energy(self):
        params = np.concatenate([[float(self.nbasis)], self.weights.astype(np.float64).ravel(),
                                 self.nu.astype(np.float64), self.bias.astype(np.float64)])
        return (_lib.E_PRODUCT_OF_T, params)

    def gen_init_X(self):
        from scipy import stats
        Zinit = np.zeros((self.ndims, self.nbatch))
        for ii in range(self.ndims):
            Zinit[ii] = stats.t.rvs(self.nu[ii], size=self.nbatch)
        Yinit = Zinit - self.bias.reshape((-1, 1))
        self.Xinit = np.dot(np.linalg.inv(self.weights), Yinit)

    def __hash__(self):
        return hash((self.ndims, self.nbasis, hash(tuple(self.nu)), hash(tuple(self.weights.ravel())),
                     hash(tuple(self.bias.ravel()))))


class SparseImageCode(Distribution):
    """Posterior over sparse-coding coefficients (mjhmc/misc/tf_distributions.py:204-284):
    E = mean_p 1/2 |y_p - B a_p|^2 + lambda * sum log(1 + a^2)   (Cauchy; ``cauchy=False``: lambda * sum |a|).

    The reference reads ``distr_data/dump_{n_basis}.pkl`` (basis (256, n_basis) and image patches), which is
    not part of the reference checkout; pass ``basis`` (img_size, n_coeffs) and ``imgs`` (img_size, >= n_patches)
    instead.  State rows are patch-major (row p * n_coeffs + c), which is what the reference's reshape yields for
    one active column.  The device kernels (bf16 matrix-core operands, fp32 accumulation) cover img_size = 256,
    n_coeffs = 1024 or 512, n_patches 1 ... 32.

    ``state_dtype``: 'float32' (default) keeps the state as the reference does (TensorFlow float32 placeholders,
    tf_distributions.py:89); 'bfloat16' is BASELINE.json configs[4] ("bf16 state / fp32 accumulate"): half the state bytes,
    the same kernel time -- and a MarkovJumpHMC chain that runs measurably hot, because a state rounded to 8 bits at every
    commit makes L irreversible (F L F L z != z at the 2^-9 level) while the jump process relies on exactly that identity
    (tests/test_gpu_stationary.py::test_sic_stationary_law).  The discrete-time samplers
    (ControlHMC / HMC: Metropolis accept) keep the law with either."""

    def __init__(self, n_patches=9, n_batches=10, cauchy=True, n_basis=1024, basis=None, imgs=None, init=None,
                 state_dtype='float32'):
        self.max_n_particles = 50
        self.lmbda = 0.01
        if basis is None or imgs is None:
            raise IOError('distr_data/dump_%d.pkl is not shipped with the reference: pass basis= and imgs=' % n_basis)
        self.basis = np.asarray(basis, dtype=np.float64)
        self.imgs = np.asarray(imgs, dtype=np.float64)
        self.img_size, self.n_coeffs = self.basis.shape
        assert self.n_coeffs == n_basis
        self.n_patches = n_patches
        self.cauchy = cauchy
        self.patches = self.imgs[:, :n_patches].T            # (n_patches, img_size)
        self._init = init
        if state_dtype not in ('float32', 'bfloat16'):
            raise ValueError("SparseImageCode state_dtype must be 'float32' or 'bfloat16'")
        self.state_dtype = state_dtype
        self.backend = 'hip-mfma-bf16'
        super(SparseImageCode, self).__init__(ndims=n_patches * self.n_coeffs, nbatch=n_batches)

    def device_energy(self):
        params = np.concatenate([[float(self.n_patches), float(self.img_size), float(self.n_coeffs), self.lmbda,
                                  1.0 if self.cauchy else 0.0], self.basis.ravel(), self.patches.ravel()])
        return (_lib.E_SPARSE_CODE, params)

    def gen_init_X(self):
        if self._init is not None:
            self.Xinit = np.asarray(self._init, dtype=np.float64)
        else:
            self.Xinit = 0.1 * np.random.randn(self.ndims, self.nbatch)

    def __hash__(self):
        return hash((hash(self.imgs.tobytes()), hash(self.basis.tobytes()), hash(self.lmbda), hash(self.n_patches),
                     self.n_coeffs))
