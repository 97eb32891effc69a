"""TEST INFRASTRUCTURE ONLY -- CPU (NumPy) restatement of the MJHMC particle hot path.

This is the parity oracle for the HIP engine in mjhmc_amd/.  It restates, operation for
operation, what the reference rueberger/MJHMC does on its NumPy path so that results are
bit-identical to the reference when fed the same random numbers:

    particle state + leapfrog operators .... mjhmc/samplers/hmc_state.py:13-148
    waiting-time draws / arg-min ........... mjhmc/misc/utils.py:15-49
    samplers (MJHMC, control variants) ..... mjhmc/samplers/markov_jump_hmc.py:16-415
    energy models .......................... mjhmc/misc/distributions.py:256-453,
                                             mjhmc/misc/tf_distributions.py:142-284

PARITY STATUS
  * pinned: everything on the NumPy path (state operators, jump bookkeeping, resampling,
    TestGaussian/Gaussian/RoughWell/MultimodalGaussian energies) -- checked bit-for-bit
    against golden vectors captured from the imported reference (oracle/capture_golden.py ->
    tests/golden/*.npz, verified by tests/test_oracle_golden.py).
  * parity unpinned by the reference: ProductOfT (Theano), Funnel and SparseImageCode (TensorFlow 0.x)
    cannot be imported anywhere (neither package is installable here; the reference pins no versions).
    Their formulas below are restated from the cited lines; the hand-derived gradients are checked
    against an independent torch restatement of the reference's forward graphs with autograd gradients
    (oracle/autograd_energies.py, tests/test_oracle_autograd.py, fixtures tests/golden/g2_dense.npz).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (mjhmc_amd) never does.

Structure note: the copies, fancy-index gathers, the per-particle Python loop in the waiting-time
draws and the *full* cached inverse-L state are kept on purpose.  They fix NumPy's summation
order (a gathered block ``A[:, idx]`` comes back F-ordered, so ``sum(axis=0)`` is a pairwise sum
along the particle's own dims) and they make this file an honest stand-in for the reference when
it is timed as the CPU baseline on a box the reference cannot travel to.
"""
from __future__ import division

import numpy as np


# --------------------------------------------------------------------------------------------
# random-number sources
# --------------------------------------------------------------------------------------------

class NonFiniteRate(ValueError):
    """Raised where the reference raises ValueError in draw_from (utils.py:41-48)."""


class GlobalNumpyRNG(object):
    """Consumes np.random.* in exactly the reference's order (seed with np.random.seed)."""

    def normals(self, ndims, n):                       # hmc_state.py:26,125
        return np.random.randn(ndims, n)

    initial_normals = normals

    def unit_exponential(self, kind, particle):        # utils.py:42 ; exponential(s) == s*std_exp()
        return np.random.standard_exponential()

    def uniforms(self, n):                             # markov_jump_hmc.py:125,132,326
        return np.random.rand(n)

    def uniform(self):                                 # markov_jump_hmc.py:138
        return np.random.random()

    def next_attempt(self):
        pass


class ReplayRNG(object):
    """Feeds back numbers captured from a reference run (or generated elsewhere).

    normals  : list of (ndims, n) arrays, consumed one per request
    exps     : list of (3, n) arrays of *unit* exponentials, one per jump attempt; entry
               [kind, particle] is used iff that particle's rate is non-zero
    uniforms : list of 1-D arrays / floats, consumed one per request
    """

    def __init__(self, normals=(), exps=(), uniforms=()):
        self._normals = list(normals)
        self._exps = list(exps)
        self._uniforms = list(uniforms)
        self.attempt = 0
        self.n_normals_used = 0
        self.n_uniforms_used = 0

    def normals(self, ndims, n):
        z = np.asarray(self._normals[self.n_normals_used], dtype=np.float64)
        assert z.shape == (ndims, n)
        self.n_normals_used += 1
        return z.copy()

    initial_normals = normals

    def unit_exponential(self, kind, particle):
        return float(self._exps[self.attempt][kind, particle])

    def uniforms(self, n):
        u = np.asarray(self._uniforms[self.n_uniforms_used], dtype=np.float64).reshape(-1)
        assert u.shape == (n,)
        self.n_uniforms_used += 1
        return u.copy()

    def uniform(self):
        u = float(np.asarray(self._uniforms[self.n_uniforms_used]).reshape(()))
        self.n_uniforms_used += 1
        return u

    def next_attempt(self):
        self.attempt += 1


class PhiloxRNG(object):
    """The engine's production RNG (oracle/philox.py), keyed by global particle id.

    ``tick`` advances once per jump attempt, so a halved-step retry sees fresh numbers.
    tick 0 is the initial momentum.
    """

    def __init__(self, seed, particle_ids):
        from .philox import PhiloxStream
        self.stream = PhiloxStream(seed, particle_ids)
        self.tick = 1
        self._exp_cache = None
        self._exp_tick = -1

    def initial_normals(self, ndims, n):
        assert n == self.stream.pid.shape[0]
        return self.stream.normals(ndims, 0)

    def normals(self, ndims, n):
        assert n == self.stream.pid.shape[0]
        return self.stream.normals(ndims, self.tick)

    def unit_exponential(self, kind, particle):
        if self._exp_tick != self.tick:
            self._exp_cache = self.stream.unit_exponentials(self.tick)
            self._exp_tick = self.tick
        return float(self._exp_cache[kind, particle])

    def uniforms(self, n):
        """1st request of a tick: accept uniforms; 2nd: flip uniforms (markov_jump_hmc.py:125,132)."""
        assert n == self.stream.pid.shape[0]
        self._ucalls = getattr(self, '_ucalls', 0) + 1
        return self.stream.accept_uniforms(self.tick) if self._ucalls == 1 else self.stream.flip_uniforms(self.tick)

    def uniform(self):
        """Batch-wide R gate (markov_jump_hmc.py:138): one number per tick, particle id 0xFFFFFFFF."""
        from .philox import PhiloxStream, SLOT_FLIP, philox4x32_10, u53
        import numpy as _np
        w = philox4x32_10(_np.array([0xFFFFFFFF], dtype=_np.uint64), self.tick & 0xFFFFFFFF, (self.tick >> 32) & 0xFFFFFFFF,
                          SLOT_FLIP, self.stream.k0, self.stream.k1)
        return float(u53(w[2], w[3])[0])

    def next_attempt(self):
        self.tick += 1
        self._ucalls = 0


# --------------------------------------------------------------------------------------------
# energy models
# --------------------------------------------------------------------------------------------

class Energy(object):
    """Counting facade, as Distribution.E / dEdX (distributions.py:62-75)."""

    def __init__(self):
        self.E_count = 0
        self.dEdX_count = 0

    def E(self, X):
        self.E_count += X.shape[1]
        return self.E_val(X)

    def dEdX(self, X):
        self.dEdX_count += X.shape[1]
        return self.dEdX_val(X)


class IsoGaussian(Energy):
    """TestGaussian (distributions.py:348-362) == the README example (README.md:18-24) at sigma=1."""

    def __init__(self, sigma=1.):
        Energy.__init__(self)
        self.sigma = sigma

    def E_val(self, X):
        return np.sum(X ** 2, axis=0).reshape((1, -1)) / (2. * self.sigma ** 2)

    def dEdX_val(self, X):
        return X / self.sigma ** 2


class DiagGaussian(Energy):
    """'Ill conditioned Gaussian' with J = diag(10**linspace(-c, 0, D)) (distributions.py:256-273).

    The dense products with the diagonal J are kept: they are what the reference evaluates.
    """

    def __init__(self, ndims=2, log_conditioning=6):
        Energy.__init__(self)
        self.conditioning = 10 ** np.linspace(-log_conditioning, 0, ndims)
        self.J = np.diag(self.conditioning)

    def E_val(self, X):
        return np.sum(X * np.dot(self.J, X), axis=0).reshape((1, -1)) / 2.

    def dEdX_val(self, X):
        return np.dot(self.J, X) / 2. + np.dot(self.J.T, X) / 2.


class RoughWell(Energy):
    """distributions.py:283-304."""

    def __init__(self, scale1=100, scale2=4):
        Energy.__init__(self)
        self.scale1 = scale1
        self.scale2 = scale2

    def E_val(self, X):
        c = np.cos(X * 2 * np.pi / self.scale2)
        return np.sum((X ** 2) / (2 * self.scale1 ** 2) + c, axis=0).reshape((1, -1))

    def dEdX_val(self, X):
        s = np.sin(X * 2 * np.pi / self.scale2)
        return X / self.scale1 ** 2 + -s * 2 * np.pi / self.scale2


class MultimodalGaussian(Energy):
    """Two unit-ish Gaussians separated along dim 0 (distributions.py:314-335).

    As coded the separation vector has 2*separation in row 0 and zeros elsewhere.
    """

    def __init__(self, ndims=2, separation=3):
        Energy.__init__(self)
        self.sep_col = np.zeros((ndims, 1))
        self.sep_col[0, 0] = 2 * separation

    def E_val(self, X):
        S = np.repeat(self.sep_col, X.shape[1], axis=1).astype(np.int64)
        return -np.log(np.exp(-np.sum((X + S) ** 2, axis=0)) + np.exp(-np.sum((X - S) ** 2, axis=0)))

    def dEdX_val(self, X):
        S = np.repeat(self.sep_col, X.shape[1], axis=1).astype(np.int64)
        common = np.exp(np.sum(4 * S * X, axis=0))
        return (2 * ((X - S) * common + S + X)) / (common + 1)


class ProductOfT(Energy):
    """Product of Student-t experts (distributions.py:420-433); gradient derived by hand where the
    reference uses Theano autodiff (:408-415).  PARITY UNPINNED (Theano not importable).

        u = (W^T x + b) / nu ;  E = sum_j (nu_j+1)/2 * log(1 + u_j^2)
        dE/dx = W . ( (nu_j+1)/2 * 2 u_j / (1 + u_j^2) / nu_j )

    ``force_dtype`` float32 mimics ``allow_input_downcast=True`` with float32 shared parameters
    (:398-415): the force is evaluated in fp32 while the integrator state stays fp64.
    """

    def __init__(self, W, lognu=None, b=None, nu=None, force_dtype=np.float64):
        Energy.__init__(self)
        self.ft = np.dtype(force_dtype)
        self.W = np.array(W, dtype=np.float32).astype(self.ft)
        if nu is None:
            nu = np.exp(lognu)
        self.nu = np.array(nu, dtype=np.float32).astype(self.ft)
        self.b = np.zeros(self.W.shape[1], dtype=self.ft) if b is None else np.array(b, dtype=np.float32).astype(self.ft)

    def _u(self, X):
        return (np.dot(self.W.T, X.astype(self.ft)) + self.b[:, None]) / self.nu[:, None]

    def E_val(self, X):
        u = self._u(X)
        alpha = (self.nu[:, None] + self.ft.type(1.)) / self.ft.type(2.)
        return np.sum(alpha * np.log(self.ft.type(1) + u ** 2), axis=0).reshape((1, -1))

    def dEdX_val(self, X):
        u = self._u(X)
        alpha = (self.nu[:, None] + self.ft.type(1.)) / self.ft.type(2.)
        h = alpha * (self.ft.type(2) * u / (self.ft.type(1) + u ** 2)) / self.nu[:, None]
        return np.dot(self.W, h)


class FunnelLiteral(Energy):
    """Funnel exactly as coded (tf_distributions.py:157-165): the *negated*, un-normalised
    log-density, E = -(D-1) x0^2/scale^2 - sum_k x_k^2 exp(-x0).  PARITY UNPINNED."""

    def __init__(self, scale=1.0):
        Energy.__init__(self)
        self.scale = float(scale)

    def E_val(self, X):
        e0 = -((X[0, :] ** 2) / (self.scale ** 2))
        ek = -((X[1:, :] ** 2) / np.exp(X[0, :]))
        return np.sum(e0 + ek, axis=0)

    def dEdX_val(self, X):
        D = X.shape[0]
        ex = np.exp(-X[0, :])
        g = np.empty_like(X)
        g[0, :] = -2. * (D - 1) * X[0, :] / self.scale ** 2 + ex * np.sum(X[1:, :] ** 2, axis=0)
        g[1:, :] = -2. * X[1:, :] * ex
        return g


class FunnelNeal(Energy):
    """Neal (2003) funnel as the reference's docstring intends (tf_distributions.py:143-147):
    x0 ~ N(0, scale^2), x_k ~ N(0, e^{x0}).  E = x0^2/(2 s^2) + sum_k x_k^2 / (2 e^{x0}) + (D-1) x0/2."""

    def __init__(self, scale=3.0):
        Energy.__init__(self)
        self.scale = float(scale)

    def E_val(self, X):
        D = X.shape[0]
        ex = np.exp(-X[0, :])
        return (X[0, :] ** 2 / (2. * self.scale ** 2) + 0.5 * ex * np.sum(X[1:, :] ** 2, axis=0)
                + 0.5 * (D - 1) * X[0, :]).reshape((1, -1))

    def dEdX_val(self, X):
        D = X.shape[0]
        ex = np.exp(-X[0, :])
        g = np.empty_like(X)
        g[0, :] = X[0, :] / self.scale ** 2 - 0.5 * ex * np.sum(X[1:, :] ** 2, axis=0) + 0.5 * (D - 1)
        g[1:, :] = X[1:, :] * ex
        return g


class SparseImageCode(Energy):
    """Sparse-coding posterior over coefficients, the per-particle maths tf_distributions.py:241-272
    intends: E = mean_p 1/2 |y_p - B a_p|^2 + lambda * sum log(1+a^2) (Cauchy) or lambda*sum|a|.
    State rows are patch-major: row p*n_coeffs + c.  PARITY UNPINNED.

    ``operand_rounding`` (a callable, e.g. round-to-bfloat16) restates the mixed-precision form BASELINE.json
    configs[4] asks for -- "bf16 state / fp32 accumulate": both matrix products take rounded operands (the
    dictionary, the coefficients entering B a, the residual entering B^T r) and accumulate exactly; everything
    else (prior, sums of squares, integrator) stays in full precision."""

    def __init__(self, basis, patches, lmbda=0.01, cauchy=True, operand_rounding=None):
        Energy.__init__(self)
        self.rnd = operand_rounding or (lambda a: a)
        self.B = self.rnd(np.asarray(basis, dtype=np.float64))  # (img, n_coeffs)
        self.Y = np.asarray(patches, dtype=np.float64)          # (n_patches, img)
        self.lmbda = lmbda
        self.cauchy = cauchy

    def _resid(self, X):
        P = self.Y.shape[0]
        C = self.B.shape[1]
        A = self.rnd(X).reshape(P, C, -1)
        recon = np.einsum('ic,pcn->pin', self.B, A)
        return recon - self.Y[:, :, None], A

    def E_val(self, X):
        R, _ = self._resid(X)
        rec = np.mean(np.sum(0.5 * R ** 2, axis=1), axis=0)
        pen = np.sum(np.log(1 + X ** 2), axis=0) if self.cauchy else np.sum(np.abs(X), axis=0)
        return (rec + self.lmbda * pen).reshape((1, -1))

    def dEdX_val(self, X):
        R, _ = self._resid(X)
        P = self.Y.shape[0]
        g = np.einsum('ic,pin->pcn', self.B, self.rnd(R / P)).reshape(X.shape)   # d/da_p of the MEAN over patches
        pen = 2 * X / (1 + X ** 2) if self.cauchy else np.sign(X)
        return g + self.lmbda * pen


class LambdaEnergy(Energy):
    """User callables, per the README contract (README.md:27-36)."""

    def __init__(self, energy_func, energy_grad_func):
        Energy.__init__(self)
        self.E_val = energy_func
        self.dEdX_val = energy_grad_func


# --------------------------------------------------------------------------------------------
# particle state and its operators (hmc_state.py)
# --------------------------------------------------------------------------------------------

class Particles(object):
    """X, V, dEdX (ndims, n); EX, EV (1, n).  The master copy also owns ``shadow`` -- the cached
    inverse-L proposal -- and ``shadow_ok`` (hmc_state.py:41-44)."""

    def __init__(self, owner, X, V=None, EX=None, EV=None, dEdX=None, is_shadow=False):
        self.owner = owner
        self.X = X
        self.n = X.shape[1]
        self.live = np.arange(self.n)
        self.V = owner.rng.initial_normals(X.shape[0], self.n) if V is None else V   # :24-26
        if EX is None:
            self.EX = np.zeros((1, self.n))
            self.refresh_EX()
        else:
            self.EX = EX
        if EV is None:
            self.EV = np.zeros((1, self.n))
            self.refresh_EV()
        else:
            self.EV = EV
        if dEdX is None:
            self.dEdX = np.zeros(X.shape)
            self.refresh_grad()
        else:
            self.dEdX = dEdX
        if not is_shadow:
            self.shadow = self.clone(as_shadow=True)
            self.shadow_ok = np.zeros(self.n, dtype=bool)

    # -- cached quantities, evaluated on the gathered live columns (:46-53)
    def refresh_EX(self):
        self.EX[:, self.live] = self.owner.E(self.X[:, self.live]).reshape((1, -1))

    def refresh_EV(self):
        self.EV[:, self.live] = np.sum(self.V[:, self.live] ** 2, axis=0).reshape((1, -1)) / 2.

    def refresh_grad(self):
        self.dEdX[:, self.live] = self.owner.dEdX(self.X[:, self.live])

    def clone(self, as_shadow=False):                                             # :55-61
        Z = Particles(self.owner, self.X.copy(), V=self.V.copy(), EX=self.EX.copy(), EV=self.EV.copy(),
                      dEdX=self.dEdX.copy(), is_shadow=as_shadow)
        Z.live = self.live.copy()
        if not as_shadow:
            Z.shadow = self.shadow.clone(True)
            Z.shadow_ok = self.shadow_ok.copy()
        return Z

    def overwrite(self, idx, Z):                                                  # :63-72
        if len(idx) == 0:
            return
        self.X[:, idx] = Z.X[:, idx]
        self.V[:, idx] = Z.V[:, idx]
        self.EX[:, idx] = Z.EX[:, idx]
        self.EV[:, idx] = Z.EV[:, idx]
        self.dEdX[:, idx] = Z.dEdX[:, idx]

    def H(self):                                                                  # :80-84
        return self.EX + self.EV

    def leap(self):                                                               # :86-91
        eps = self.owner.epsilon
        self.V[:, self.live] += -eps / 2. * self.dEdX[:, self.live]
        self.X[:, self.live] += eps * self.V[:, self.live]
        self.refresh_grad()
        self.V[:, self.live] += -eps / 2. * self.dEdX[:, self.live]

    def L(self):                                                                  # :93-100
        for _ in range(self.owner.num_leapfrog_steps):
            self.leap()
        rnd = getattr(self.owner, 'state_rounding', None)
        if rnd is not None:      # reduced-precision STATE (bf16 / float32 configs): the end point is stored rounded
            self.X[:, self.live] = rnd(self.X[:, self.live])
            self.V[:, self.live] = rnd(self.V[:, self.live])
        self.refresh_EV()
        self.refresh_EX()
        return self

    def F(self):                                                                  # :102-107
        self.V[:, self.live] = -self.V[:, self.live]
        return self

    def FLF(self):                                                                # :109-119
        warm = np.where(self.shadow_ok == True)[0]       # noqa: E712
        self.live = np.where(self.shadow_ok == False)[0]  # noqa: E712
        Z = self.F().L().F()
        Z.overwrite(warm, self.shadow)
        self.live = np.arange(self.n)
        return Z

    def R(self):                                                                  # :121-129
        beta = self.owner.beta
        self.V = self.V * np.sqrt(1. - beta) + self.owner.rng.normals(self.owner.ndims, self.n) * np.sqrt(beta)
        rnd = getattr(self.owner, 'state_rounding', None)
        if rnd is not None:
            self.V = rnd(self.V)
        self.refresh_EV()
        return self

    def remember_flf(self, idx, Z):                                               # :131-136
        self.shadow.overwrite(idx, Z)
        self.shadow_ok[idx] = True

    def forget_flf(self, idx):                                                    # :138-143
        self.shadow_ok[idx] = False

    def wipe_flf(self):                                                           # :145-148
        self.shadow_ok = np.zeros_like(self.shadow_ok)


# --------------------------------------------------------------------------------------------
# waiting times and arg-min (utils.py)
# --------------------------------------------------------------------------------------------

def waiting_times(rates, rng, kind):
    """utils.py:31-49 -- one Exp(rate) draw per particle, in particle order.

    rate == 0 -> inf without consuming a number; non-finite rate -> NonFiniteRate at that particle
    (numbers for earlier particles have been consumed, as in the reference's loop)."""
    assert rates.ndim == 1
    out = []
    for i, rate in enumerate(rates):
        if rate == 0:
            out.append(np.inf)
        elif np.isfinite(rate):
            out.append((1. / rate) * rng.unit_exponential(kind, i))
        else:
            raise NonFiniteRate("non-finite transition rate at particle %d" % i)
    return np.array(out).reshape(1, len(rates))


def first_minimum(draws):
    """utils.py:15-28 -- index lists of the particles whose minimum is row k (ties -> lowest row)."""
    stacked = np.concatenate(draws, axis=0)
    which = np.argmin(stacked, axis=0)
    return [np.where(which == k)[0] for k in range(len(draws))], which


# --------------------------------------------------------------------------------------------
# samplers (markov_jump_hmc.py)
# --------------------------------------------------------------------------------------------

class _SamplerCore(object):
    """Hyper-parameters, counters, E/dEdX plumbing (markov_jump_hmc.py:67-104)."""

    def __init__(self, energy, Xinit, epsilon=1e-4, alpha=0.2, beta=None, num_leapfrog_steps=5,
                 rng=None, V0=None, build_state=True, state_rounding=None):
        self.rng = GlobalNumpyRNG() if rng is None else rng
        self.state_rounding = state_rounding     # None on the reference's float64 path (every golden fixture)
        self.energy = energy
        self.ndims, self.nbatch = Xinit.shape
        self.num_leapfrog_steps = num_leapfrog_steps
        self.epsilon = epsilon
        self.beta = beta or alpha ** (1. / (self.epsilon * self.num_leapfrog_steps))       # :69
        self.original_epsilon = epsilon
        self.original_l = num_leapfrog_steps
        self.n_burn_in = 500
        self.p_flip = 0.5
        self.p_r = 1
        self.l_count = 0
        self.f_count = 0
        self.fl_count = 0
        self.r_count = 0
        self.grad_per_sample_step = self.num_leapfrog_steps
        self._V0 = V0
        if build_state:
            self.state = Particles(self, Xinit.copy(), V=None if V0 is None else V0.copy())

    def E(self, X):                                                                        # :95-98
        return self.energy.E(X).reshape((1, -1))

    def dEdX(self, X):                                                                     # :101-104
        return self.energy.dEdX(X)

    def burn_in(self):                                                                     # :176-180
        for _ in range(self.n_burn_in):
            self.sampling_iteration()


class HMCBase(_SamplerCore):
    """Discrete-time control sampler: FL proposal, MH accept, coin-flip F, batch-wide R (:116-148)."""

    def accept_prob(self, Z1, Z2):                                                         # :106-114
        d = Z1.H() - Z2.H()
        p = np.ones((1, d.shape[1]))
        p[d < 0] = np.exp(d[d < 0])
        return p

    def sampling_iteration(self):
        proposal = self.state.clone().L().F()
        p_acc = self.accept_prob(self.state, proposal)
        everyone = np.arange(self.nbatch).reshape(1, self.nbatch)
        fl_idx = everyone[self.rng.uniforms(self.nbatch) < p_acc]
        self.state.overwrite(fl_idx, proposal)
        p_half = self.p_flip * np.ones((1, self.nbatch))
        flip_idx = everyone[self.rng.uniforms(self.nbatch) < p_half]
        flipped = self.state.clone().F()
        self.state.overwrite(flip_idx, flipped)
        if self.rng.uniform() < self.p_r:
            self.r_count += self.nbatch
            self.state.R()
        moved, flips = set(fl_idx), set(flip_idx)
        self.l_count += len(moved & flips)
        self.f_count += len(flips - moved)
        self.fl_count += len(moved - flips)
        self.last_fl_idx, self.last_flip_idx = fl_idx, flip_idx
        self.rng.next_attempt()

    def sample(self, n_samples=1000, preserve_order=False):                                # :150-173
        out = []
        for _ in range(n_samples):
            self.sampling_iteration()
            out.append(self.state.clone().X)
        return np.stack(out, axis=-1) if preserve_order else np.concatenate(out, axis=1)


class HMC(HMCBase):                                                                        # :183-189
    def __init__(self, *a, **k):
        HMCBase.__init__(self, *a, **k)
        self.p_flip = 1


class ControlHMC(HMCBase):                                                                 # :191-200
    def __init__(self, *a, **k):
        HMCBase.__init__(self, *a, **k)
        self.p_flip = 1
        self.p_r = - np.log(1 - self.beta) * 0.5
        self.beta = 1


class ContinuousTimeHMC(HMCBase):
    """F / FL / R jump process with unit flip rate (:203-347)."""

    def __init__(self, energy, Xinit, resample=True, **k):
        self.resample = resample
        HMCBase.__init__(self, energy, Xinit, build_state=False, **k)
        self.p_r = - np.log(1 - self.beta) * 0.5                                           # :221
        self.beta = 1                                                                      # :223
        self.state = Particles(self, Xinit.copy(), V=None if self._V0 is None else self._V0.copy())
        self.dwelling_times = np.zeros(self.nbatch)

    def transition_rates(self, Z1, Z2):                                                    # :341-347
        return np.exp(Z1.H() - Z2.H()) ** .5

    def sampling_iteration(self):                                                          # :251-290
        f_state = self.state.clone().F()
        fl_state = self.state.clone().L().F()
        fl_rates = self.transition_rates(self.state, fl_state)
        f_rates = np.ones((1, self.nbatch))
        r_rates = self.p_r * np.ones((1, self.nbatch))
        fl_draws = waiting_times(fl_rates[0], self.rng, 0)
        f_draws = waiting_times(f_rates[0], self.rng, 1)
        r_draws = waiting_times(r_rates[0], self.rng, 2)
        (f_idx, fl_idx, r_idx), _ = first_minimum([f_draws, fl_draws, r_draws])
        self.dwelling_times = np.amin(np.concatenate((fl_draws, f_draws, r_draws)), axis=0)
        self.state.overwrite(fl_idx, fl_state)
        self.state.overwrite(f_idx, f_state)
        r_state = self.state.clone().R()
        self.state.overwrite(r_idx, r_state)
        self.fl_count += len(fl_idx)
        self.f_count += len(f_idx)
        self.r_count += len(r_idx)
        self.rng.next_attempt()

    def sample(self, n_samples=1000, preserve_order=False):                                # :293-338
        if self.resample:
            kept, dwell = [], []
            self.sampling_iteration()
            kept.append(self.state.clone().X)
            for _ in range(n_samples):
                dwell.append(self.dwelling_times.copy())
                self.sampling_iteration()
                kept.append(self.state.clone().X)
            dwell_t = np.concatenate(dwell)
            pool = np.concatenate(kept[:-1], axis=1)
            total_t = np.sum(dwell_t)
            cumul_t = np.cumsum(dwell_t)
            u = np.sort(self.rng.uniforms(n_samples * self.nbatch)) * total_t
            # first index with cumul_t > u  ==  the reference's np.where(...)[0][0] loop (:326-328)
            pick = np.searchsorted(cumul_t, u, side='right')
            self.last_pick = pick
            return pool[:, pick]
        out = []
        for _ in range(n_samples):
            self.sampling_iteration()
            out.append(self.state.clone().X)
        return np.stack(out, axis=-1) if preserve_order else np.concatenate(out, axis=1)


class MarkovJumpHMC(ContinuousTimeHMC):
    """The L / F / R jump process with the cached inverse-L proposal (:350-415)."""

    def sampling_iteration(self):
        f_state = self.state.clone().F()
        l_state = self.state.clone().L()
        flf_state = self.state.clone().FLF()
        r_state = self.state.clone().R()
        try:
            l_rates = self.transition_rates(self.state, l_state)
            flf_rates = self.transition_rates(self.state, flf_state)
            f_rates = flf_rates - np.min((flf_rates, l_rates), axis=0)
            r_rates = self.p_r * np.ones((1, self.nbatch))
            try:
                l_draws = waiting_times(l_rates[0], self.rng, 0)
                f_draws = waiting_times(f_rates[0], self.rng, 1)
                r_draws = waiting_times(r_rates[0], self.rng, 2)
            finally:
                self.rng.next_attempt()
        except NonFiniteRate:                                                              # :376-389
            self.epsilon *= 0.5
            self.num_leapfrog_steps *= 2
            self.retry_depths = getattr(self, 'retry_depths', [])
            self.retry_depths.append(np.log(self.original_epsilon / self.epsilon) / np.log(2))
            self.state.wipe_flf()
            self.sampling_iteration()
            self.epsilon *= 2
            self.num_leapfrog_steps = int(self.num_leapfrog_steps / 2)
            return
        (l_idx, f_idx, r_idx), which = first_minimum([l_draws, f_draws, r_draws])
        self.dwelling_times = np.amin(np.concatenate((l_draws, f_draws, r_draws)), axis=0)
        self.state.remember_flf(l_idx, self.state)
        self.state.overwrite(l_idx, l_state)
        self.state.overwrite(f_idx, f_state)
        self.state.overwrite(r_idx, r_state)
        self.state.forget_flf(r_idx)
        self.state.forget_flf(f_idx)
        self.l_count += len(l_idx)
        self.f_count += len(f_idx)
        self.r_count += len(r_idx)
        self.last_transition = which.astype(np.uint8)
