"""GPU, counter RNG: distribution-level checks of the energies the reference builds symbolically (ProductOfT: Theano,
Funnel: TensorFlow 0.x) and that therefore cannot be pinned by running the reference.  These tests use NO oracle and no
restated algebra -- only facts that follow from the reference's own formulas:

ProductOfT, E = sum_j (nu_j + 1)/2 * log(1 + ((W^T x + b)_j / nu_j)^2)   (mjhmc/misc/distributions.py:420-433).
    With W invertible the change of variables y = W^T x + b has a constant Jacobian and the density factorises:
    p(y_j) ~ (1 + y_j^2 / nu_j^2)^(-(nu_j + 1)/2), i.e.  y_j / sqrt(nu_j) ~ Student-t(nu_j), independently.
    (The reference's own gen_init_X, :437-445, draws Student-t(nu_j) for y_j itself and maps through inv(W), not
    inv(W^T): it is not the law of this energy; the negative controls below show the test tells the two apart.)
Funnel as documented, x_0 ~ N(0, scale^2), x_k | x_0 ~ N(0, e^{x_0})   (mjhmc/misc/tf_distributions.py:143-147,171-173):
    x_0 / scale ~ N(0, 1) and x_k * e^{-x_0 / 2} ~ N(0, 1).

A sampler whose energy, accept / jump rule or momentum refresh were wrong would not leave these laws invariant: chains
started FROM the law must still follow it after many iterations, and chains started away from it must arrive there.
The GRADIENT is not tested by that (any reversible volume-preserving proposal keeps the law): it is tested by energy
conservation -- the leapfrog error |H(L z) - H(z)| must be small and shrink 4x when epsilon halves -- with a negative
control in which the gradient is deliberately doubled.

Thresholds: Kolmogorov-Smirnov p-values at fixed seeds, Bonferroni-style: every coordinate's p > 1e-4 (36 or 10
coordinates: a correct sampler fails a run with probability < 0.4 %), negative controls p < 1e-6 (median over the
coordinates).
"""
import numpy as np
import pytest
from scipy import stats

from tests.helpers import ref_init_weights

pytestmark = pytest.mark.gpu
P_MIN = 1e-4


def _pot_model(D=36):
    W, lognu = ref_init_weights(D, D)          # the reference recipe (search/MJHMC_poe_36/mjhmc_objective.py:15-23)
    W = (W + np.eye(D)).astype(np.float32).astype(np.float64)
    assert np.linalg.cond(W) < 1e4
    nu = np.exp(lognu).astype(np.float32).astype(np.float64)
    return W, lognu, nu


def _pot_exact_draw(W, nu, N, rs):
    t = np.stack([rs.standard_t(nu[j], size=N) for j in range(len(nu))])
    return np.linalg.solve(W.T, t * np.sqrt(nu)[:, None])       # x = W^-T y, b = 0


def _pot_pvalues(W, nu, X, df_scale=1.0, y_scale=None):
    y = W.T.dot(X)
    t = y / (np.sqrt(nu)[:, None] if y_scale is None else y_scale)
    return np.array([stats.kstest(t[j], 't', args=(nu[j] * df_scale,)).pvalue for j in range(len(nu))])


def _run(cls_name, dist, n_iter, **kw):
    from mjhmc_amd.samplers import markov_jump_hmc as M
    s = getattr(M, cls_name)(distribution=dist, **kw)
    for _ in range(n_iter):
        s.sampling_iteration()
    return s


@pytest.mark.parametrize('cls_name,state', [('MarkovJumpHMC', 'float32'), ('ControlHMC', 'float32'),
                                            ('MarkovJumpHMC', 'float64'), ('ControlHMC', 'float64')])
def test_product_of_t_stationary_law(cls_name, state):
    """`state`: float32 state and force (dense_pot.hip) / the reference's arithmetic, float64 state around the float32 force
    (dense_pot64.hip: the tile kernel with the state streamed through its epilogue)."""
    from mjhmc_amd.misc.distributions import ProductOfT
    D, N = 36, 8192
    W, lognu, nu = _pot_model(D)
    rs = np.random.RandomState(2026)
    X_exact = _pot_exact_draw(W, nu, N, rs)
    assert _pot_pvalues(W, nu, X_exact).min() > P_MIN                         # the test statistic on an exact draw

    def dist_from(X0):
        class Fixed(ProductOfT):
            def gen_init_X(self):
                self.Xinit = X0
        return Fixed(ndims=D, nbasis=D, nbatch=N, lognu=lognu, W=W, state_dtype=state)

    kw = dict(epsilon=0.25, beta=0.3, num_leapfrog_steps=6, seed=11)
    # (1) started from the law: still the law after 80 iterations, and the chain has moved
    s = _run(cls_name, dist_from(X_exact), 80, **kw)
    X = s.state.X
    moved = np.mean(np.abs(X - X_exact) > 1e-3)
    assert moved > 0.9, moved
    p = _pot_pvalues(W, nu, X)
    assert p.min() > P_MIN, ('stationarity', p.min(), int(p.argmin()))
    # (2) started away from the law (every particle at half its distance from the mode): arrives there.  (Student-t
    # tails with 2.1 < nu < 4.1 fill slowly under Gaussian momenta: 2000 iterations.)
    X_bad = 0.5 * X_exact
    assert np.median(_pot_pvalues(W, nu, X_bad)) < 1e-6
    s2 = _run(cls_name, dist_from(X_bad), 2000, **kw)
    p2 = _pot_pvalues(W, nu, s2.state.X)
    assert p2.min() > P_MIN, ('convergence', p2.min(), int(p2.argmin()))
    # negative controls: the same samples against laws that are NOT this energy's
    assert np.median(_pot_pvalues(W, nu, X, y_scale=nu[:, None])) < 1e-6       # u_j = y_j / nu_j itself taken for Student-t(nu_j)
    assert np.median(_pot_pvalues(W, nu, X, y_scale=1.0)) < 1e-6               # y_j ~ t(nu_j) unscaled (the reference's gen_init_X)
    # the trajectories are accepted: E and dE/dX agree (ControlHMC flips every particle, p_flip = 1: accepted moves are
    # its l_count, markov_jump_hmc.py:143-148,197-200)
    acc = s.l_count / float(80 * N)
    assert acc > 0.5, acc


def _funnel_exact_draw(D, N, scale, rs):
    x0 = scale * rs.randn(1, N)
    return np.vstack([x0, np.exp(x0 / 2.) * rs.randn(D - 1, N)])


def _funnel_pvalues(X, scale):
    z = np.vstack([X[:1] / scale, X[1:] * np.exp(-X[:1] / 2.)])
    return np.array([stats.kstest(z[k], 'norm').pvalue for k in range(X.shape[0])])


@pytest.mark.parametrize('cls_name', ['MarkovJumpHMC', 'ControlHMC'])
def test_funnel_stationary_law(cls_name):
    from mjhmc_amd.misc.distributions import Funnel
    D, N, scale = 10, 8192, 1.5
    rs = np.random.RandomState(77)
    X_exact = _funnel_exact_draw(D, N, scale, rs)
    assert _funnel_pvalues(X_exact, scale).min() > P_MIN

    def dist_from(X0):
        class Fixed(Funnel):
            def gen_init_X(self):
                self.Xinit = X0
        return Fixed(scale=scale, nbatch=N, ndims=D)

    kw = dict(epsilon=0.15, beta=0.3, num_leapfrog_steps=8, seed=5)
    s = _run(cls_name, dist_from(X_exact), 120, **kw)
    X = s.state.X
    assert np.mean(np.abs(X - X_exact) > 1e-6) > 0.9
    p = _funnel_pvalues(X, scale)
    assert p.min() > P_MIN, ('stationarity', p.min(), int(p.argmin()))
    # started from a standard normal in every coordinate (x_0 too narrow, x_k not scaled by the neck): arrives at the law
    X_bad = rs.randn(D, N)
    assert _funnel_pvalues(X_bad, scale).min() < 1e-6
    s2 = _run(cls_name, dist_from(X_bad), 1500, **kw)
    p2 = _funnel_pvalues(s2.state.X, scale)
    assert p2.min() > P_MIN, ('convergence', p2.min(), int(p2.argmin()))
    # negative controls: the same samples against a funnel of another scale, and against no neck at all
    assert _funnel_pvalues(X, 2.0 * scale)[0] < 1e-6
    assert np.median([stats.kstest(X[k], 'norm').pvalue for k in range(1, D)]) < 1e-6


def _energy_error(en, X, V, eps, L, dtype):
    E0, _ = en.eval(X, want_E=True, want_grad=False, dtype=dtype)
    Xo, Vo, EX, EV, _ = en.leapfrog(X, V, eps, L, want_grad=False, dtype=dtype)
    H0 = E0 + 0.5 * np.sum(V ** 2, axis=0)
    return np.abs((EX + EV) - H0)


def test_leapfrog_conserves_energy_second_order_and_a_wrong_gradient_does_not():
    """dE/dX is the gradient of E: the leapfrog error |H(L z) - H(z)| is O(eps^2) -- small, and 4x smaller at eps / 2 --
    for ProductOfT (float32 matrix cores), the funnel (float64) and the funnel stated as coupled expressions; the SAME
    expressions with the gradient doubled (negative control) break both properties."""
    from mjhmc_amd import engine, _lib
    ctx = engine.context(0)
    rs = np.random.RandomState(3)
    # ProductOfT
    D, N = 36, 4096
    W, lognu, nu = _pot_model(D)
    en = engine.DeviceEnergy(ctx, _lib.E_PRODUCT_OF_T, D, np.concatenate([[float(D)], W.ravel(), nu, np.zeros(D)]))
    X, V = _pot_exact_draw(W, nu, N, rs), rs.randn(D, N)
    e1 = np.median(_energy_error(en, X, V, 0.1, 8, 'float32'))
    e2 = np.median(_energy_error(en, X, V, 0.05, 16, 'float32'))
    assert e1 < 0.05 and 3.0 < e1 / e2 < 5.0, (e1, e2)
    # ... and in the reference's arithmetic (float64 state around the float32 force), down to where the float32 force's
    # own error (~1e-6 of |g|) starts to show
    e1 = np.median(_energy_error(en, X, V, 0.1, 8, 'float64'))
    e2 = np.median(_energy_error(en, X, V, 0.05, 16, 'float64'))
    e3 = np.median(_energy_error(en, X, V, 0.025, 32, 'float64'))
    assert e1 < 0.05 and 3.5 < e1 / e2 < 4.5 and 3.5 < e2 / e3 < 4.5, (e1, e2, e3)
    # Funnel: built-in functor and the coupled-expression form, right and (negative control) wrong
    D, N, scale = 10, 4096, 1.5
    X, V = _funnel_exact_draw(D, N, scale, rs), rs.randn(D, N)
    good = dict(stats=['d == 0 ? x : 0.0', 'd == 0 ? 0.0 : x*x'], energy='0.0',
                energy0='S[0]*S[0]/(2*p[0]*p[0]) + 0.5*exp(-S[0])*S[1] + 0.5*(p[1]-1)*S[0]',
                grad='d == 0 ? x/(p[0]*p[0]) - 0.5*exp(-x)*S[1] + 0.5*(p[1]-1) : x*exp(-S[0])')
    wrong = dict(good, grad='2.0*(' + good['grad'] + ')')
    builtin = engine.DeviceEnergy(ctx, _lib.E_FUNNEL_NEAL, D, [scale])
    expr = engine.DeviceEnergy.from_expr(ctx, D, good['energy'], good['grad'], [scale, float(D)], good['stats'], good['energy0'])
    bad = engine.DeviceEnergy.from_expr(ctx, D, wrong['energy'], wrong['grad'], [scale, float(D)], wrong['stats'], wrong['energy0'])
    for e in (builtin, expr):
        a = np.median(_energy_error(e, X, V, 0.05, 8, 'float64'))
        b = np.median(_energy_error(e, X, V, 0.025, 16, 'float64'))
        assert a < 0.02 and 3.5 < a / b < 4.5, (a, b)
    a = np.median(_energy_error(bad, X, V, 0.05, 8, 'float64'))
    b = np.median(_energy_error(bad, X, V, 0.025, 16, 'float64'))
    assert a > 0.05 and not (3.0 < a / b < 5.0), (a, b)


@pytest.mark.parametrize('nc,P,cauchy', [(1024, 1, True), (1024, 1, False), (1024, 9, True), (1024, 9, False),
                                         (512, 1, True), (512, 1, False), (512, 9, True), (512, 9, False)])
def test_sic_leapfrog_conserves_energy_second_order(nc, P, cauchy):
    """An oracle-independent check of the bf16 SparseImageCode kernel (tf_distributions.py:241-272): its dE/dX is the
    gradient of its E.  The leapfrog energy error H(L z) - H(z) through mjhmc_leapfrog (bf16 state, bf16 matrix-core
    operands, float32 accumulation) is second order in the step: at eps / 2 and twice the steps its mean over the batch
    is ~4x smaller.  (The MEAN: rounding the end point to bf16 adds zero-mean noise of ~0.05 per particle -- the floor
    the median |dH| runs into below eps = 0.025, tools/sic_energy_error.py -- which averages out over 512 chains.)
    Negative control: the same trajectories measured with the energy of a prior twice as strong (i.e. the kernel's
    prior force is half what that H needs) -- the error is large and does not shrink."""
    from mjhmc_amd import engine, _lib
    from tests.helpers import sic_problem, to_bf16
    ctx = engine.context(0)
    B, imgs, a0 = sic_problem(3, n_patches=P, n_coeffs=nc)
    N, D = 512, P * nc
    rs = np.random.RandomState(1)
    X = to_bf16(a0[:, None] + 0.1 * rs.randn(D, N))
    V = to_bf16(rs.randn(D, N))

    def energy(lam):
        return engine.DeviceEnergy(ctx, _lib.E_SPARSE_CODE, D, np.concatenate(
            [[float(P), 256.0, float(nc), lam, 1.0 if cauchy else 0.0], B.ravel(), imgs[:, :P].T.ravel()]))
    en, en2 = energy(0.01), energy(0.02)

    def mean_dH(measure, eps, L):
        E0, _ = measure.eval(X, want_E=True, want_grad=False, dtype='bfloat16')
        Xo, Vo, EX, EV, _ = en.leapfrog(X, V, eps, L, want_grad=False, dtype='bfloat16')
        if measure is en:                                   # the operator's own energies ARE those of the stored end point
            E1, _ = en.eval(Xo, want_E=True, want_grad=False, dtype='bfloat16')
            assert np.allclose(EX, E1, rtol=1e-5, atol=1e-4) and np.allclose(EV, 0.5 * np.sum(Vo ** 2, axis=0), rtol=1e-5)
        else:
            E1, _ = measure.eval(Xo, want_E=True, want_grad=False, dtype='bfloat16')
        return float(np.mean((E1 + 0.5 * np.sum(Vo ** 2, axis=0)) - (E0 + 0.5 * np.sum(V ** 2, axis=0))))

    m1, m2 = mean_dH(en, 0.1, 4), mean_dH(en, 0.05, 8)
    assert abs(m2) < 0.25 and 3.0 < m1 / m2 < 6.5, (m1, m2)
    w1, w2 = mean_dH(en2, 0.1, 4), mean_dH(en2, 0.05, 8)
    assert abs(w2) > 0.5 and abs(w2) > 4 * abs(m2) and w1 / w2 < 2.0, (w1, w2)


@pytest.mark.parametrize('mode_name,state', [('CONTROL', 'bfloat16'), ('CONTROL', 'float32'), ('MJHMC', 'float32'),
                                             ('MJHMC', 'bfloat16')])
def test_sic_stationary_law(mode_name, state):
    """The SparseImageCode chains against the law of the energy the reference writes down (tf_distributions.py:241-272),
    with NO oracle and no second chain:
        p(a) ~ exp(-1/2 |y - B a|^2) * prod_i (1 + a_i^2)^(-lambda),   lambda = 0.01, B (256, 1024) of full row rank.
    In the 256 directions the data see, the prior is all but flat (its log-density moves by < 0.01 per unit of a): the
    residual r = y - B a is N(0, I_256) to well within the power of this test, so 1/2 |r|^2 ~ Gamma(128, 1) and every
    pixel's r_i ~ N(0, 1); the momentum is N(0, I_1024), 1/2 |v|^2 ~ Gamma(512, 1).  (In the other 768 directions the
    Cauchy prior with lambda = 0.01 is not normalisable: the chain diffuses there for ever, as the reference's does;
    nothing below looks at them.)  The chains start where the benchmark starts them -- a0 + 0.1 noise, residuals five
    times too small -- and must ARRIVE at the law with the benchmark's hyper-parameters (eps 0.05, L 25, beta 0.1).
    The state is read back and the statistics are formed on the host in float64.  A chain running 3 % too hot or too
    cold fails: the negative controls test the same samples against those laws.

    What the four cases say (round 5; tools/debug/sic_law3.py prints the numbers over time):
      * ControlHMC (Metropolis accept), bfloat16 or float32 state, and MarkovJumpHMC with float32 state (the reference's
        own state type; the class's default): the law, to ~1 % in temperature -- mean 1/2 |r|^2 = 127.9 / 129.0 / 126.6
        after 1 500 iterations, 128.2 / 130.9 / 129.5 after 3 000.  What is left is the bf16 OPERAND of the matrix cores:
        the energy the kernels evaluate is 1/2 |y - bf16(B) bf16(a)|^2, piecewise constant in a, and as the chain
        diffuses in the 768 flat directions (|a| rms 23 -> 35) its cells coarsen.  Hence a 3 % band on the mean instead
        of a Kolmogorov-Smirnov test at N = 4096 (which resolves 0.5 %); the SHAPE of the law is tested with the
        temperature scaled out.
      * MarkovJumpHMC, bfloat16 state (BASELINE.json configs[4]): NOT the law -- the chain runs hot and keeps heating
        (1/2 |r|^2 = 137 after 500 iterations, 196 after 1 500, 206 after 3 000, with or without dwell-time weighting, at
        any step size).  Rounding the state to 8 bits at every commit makes L irreversible at the 2^-9 level -- F L F L z != z,
        and the energy of a rounded end point is noisy by O(1) once |a| has diffused to ~30 in the flat directions --
        while the jump process caches H(z) as the energy of F L F (L z) and balances its rates on that identity.  The
        NumPy oracle with the same state rounding and exact float64 forces heats the same way
        (tools/debug/sic_law_oracle.py: 139 after 1 000 iterations where float64 state gives 128); a Metropolis accept
        does not care.  The case below asserts the bias so that it cannot go unnoticed in either direction."""
    from mjhmc_amd import engine, _lib
    from tests.helpers import sic_problem
    ctx = engine.context(0)
    B, imgs, a0 = sic_problem(0)
    y = imgs[:, 0]
    N, D, n_burn = 4096, 1024, 1500
    en = engine.DeviceEnergy(ctx, _lib.E_SPARSE_CODE, D, np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), y]))
    X0 = a0[:, None] + 0.1 * np.random.RandomState(12).randn(D, N)
    mode = {'MJHMC': _lib.MODE_MJHMC, 'CONTROL': _lib.MODE_CONTROL}[mode_name]
    s = engine.DeviceSampler(en, X0, seed=2027, dtype=state, mode=mode)
    beta = 0.1
    # MJHMC: refresh clock of rate p_r, full refresh share beta per R move (markov_jump_hmc.py:341-347, hmc_state.py:121-129);
    # CONTROL: the batch-wide gate fires with probability p_r (markov_jump_hmc.py:138-141)
    s.set_hparams(0.05, 25, -np.log(1 - beta) * 0.5 if mode_name == 'MJHMC' else 0.3, beta if mode_name == 'MJHMC' else 1.0, 1.0)
    r0 = y[:, None] - B.dot(X0)
    assert np.median(0.5 * np.sum(r0 ** 2, axis=0)) < 20           # the start is far from the law (128 +- 11)
    moves = np.zeros(3)
    for _ in range(n_burn // 100):
        st, done = s.iterate(100)
        assert done == 100
        moves += [sum(t.l for t in st), sum(t.f for t in st), sum(t.r for t in st)]
    X, V = s.read(_lib.F_X), s.read(_lib.F_V)
    s.close()
    r = y[:, None] - B.dot(X)
    e_data, e_kin = 0.5 * np.sum(r ** 2, axis=0), 0.5 * np.sum(V ** 2, axis=0)
    t_data, t_kin = e_data.mean() / 128.0, e_kin.mean() / 512.0          # the temperatures the chain runs at
    p_shape = stats.kstest(e_data / t_data, 'gamma', args=(128.0,)).pvalue      # the law's shape, temperature scaled out
    p_kin = stats.kstest(e_kin / t_kin, 'gamma', args=(512.0,)).pvalue
    p_pix = np.array([stats.kstest(r[i] / np.sqrt(t_data), 'norm').pvalue for i in range(0, 256, 8)])
    info = dict(mean_e_data=float(e_data.mean()), mean_e_kin=float(e_kin.mean()), lfr=(moves / moves.sum()).round(3).tolist(),
                p_shape=p_shape, p_kin=p_kin, p_pix_min=float(p_pix.min()))
    if (mode_name, state) == ('MJHMC', 'bfloat16'):
        assert t_data > 1.2 and t_kin > 1.05, ('the bf16-state jump process no longer runs hot?', info)
        return
    assert abs(t_data - 1) < 0.03 and abs(t_kin - 1) < 0.01, info
    assert p_shape > P_MIN and p_kin > P_MIN and p_pix.min() > P_MIN, info
    # negative controls: the same samples against the laws of a chain 6 % too hot / too cold
    for scale in (1.06, 0.94):
        assert stats.kstest(e_data, 'gamma', args=(128.0, 0.0, scale)).pvalue < 1e-6, (scale, info)
