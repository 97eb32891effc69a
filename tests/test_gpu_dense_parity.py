"""GPU: the energies the reference builds symbolically (ProductOfT / Funnel / SparseImageCode), through the C ABI,
against (1) the committed G2 fixtures -- torch restatement of the reference's forward graphs + autograd gradients,
oracle/capture_dense_fixtures.py -- and (2) the NumPy oracle, from the particle states the reference itself ships
(initializations/*.pickle -> tests/golden/ref_init_states.npz).

Bars.  Funnel (float64): 1e-10 relative.  ProductOfT (float32 kernel; the reference also evaluates it in float32):
the spread between the float32 and float64 fixture values times a small factor.  SparseImageCode (bf16 operands,
float32 accumulation): against the mixed-precision restatement of the oracle (operands rounded to bf16, exact
accumulation) at float32-accumulation tolerance, against the full-precision fixtures at bf16 tolerance.
Transitions: equal to the oracle's, except particles that are provably NEAR TIES (tests/helpers.py:
explainable_transitions) -- no agreement percentages.
"""
import numpy as np
import pytest

from oracle import mjhmc_oracle as orc
from tests.helpers import (load, ref_init_weights, sic_problem, to_bf16, resync as _resync, check_iteration,
                           check_control_iteration, hooks_context)

pytestmark = pytest.mark.gpu
np.seterr(all='ignore')


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


# ---------------------------------------------------------------------------------------------
# Funnel: as coded (E_FUNNEL_REF, tf_distributions.py:157-165) and as documented (E_FUNNEL_NEAL)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag,scale,literal', [('funnel_lit_s1_10x50', 1.0, True), ('funnel_lit_s3_10x50', 3.0, True),
                                               ('funnel_neal_s3_10x50', 3.0, False), ('funnel_neal_s3_32x64', 3.0, False)])
def test_funnel_single_evaluation_matches_autograd_fixture(tag, scale, literal):
    from mjhmc_amd.misc.distributions import Funnel
    g = load('g2_dense')
    X = g[tag + '_X']
    D, n = X.shape
    d = Funnel(scale=scale, nbatch=n, ndims=D, literal=literal)
    E, G = d.E(X), d.dEdX(X)
    assert E.shape == (1, n) and G.shape == (D, n) and (d.E_count, d.dEdX_count) == (n, n)
    assert rel(E[0], g[tag + '_E']) < 1e-10
    assert rel(G, g[tag + '_g']) < 1e-10
    o = (orc.FunnelLiteral if literal else orc.FunnelNeal)(scale)
    assert rel(E[0], np.asarray(o.E_val(X)).reshape(-1)) < 1e-10 and rel(G, o.dEdX_val(X)) < 1e-10


def test_funnel_literal_leapfrog_step_matches_oracle():
    """One L() of the as-coded funnel (the chains diverge on it, so one short trajectory is all that is compared)."""
    from mjhmc_amd.misc.distributions import Funnel
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    g = load('g2_dense')
    X0 = g['funnel_lit_s1_10x50_X'] * 0.3
    V0 = np.random.RandomState(1).randn(*X0.shape)

    class Fixed(Funnel):
        def gen_init_X(self):
            self.Xinit = X0

    s = MarkovJumpHMC(distribution=Fixed(scale=1.0, nbatch=50, ndims=10, literal=True), epsilon=0.01, beta=0.2,
                      num_leapfrog_steps=3, Vinit=V0, seed=1)
    Z = s.state.copy().L()
    o = orc.MarkovJumpHMC(orc.FunnelLiteral(1.0), X0, epsilon=0.01, beta=0.2, num_leapfrog_steps=3, V0=V0,
                          rng=orc.ReplayRNG())
    Zo = o.state.clone().L()
    assert rel(Z.X, Zo.X) < 1e-10 and rel(Z.V, Zo.V) < 1e-10
    assert rel(Z.EX, Zo.EX) < 1e-10 and rel(Z.EV, Zo.EV) < 1e-10


# ---------------------------------------------------------------------------------------------
# ProductOfT: single evaluations against the autograd fixtures
# ---------------------------------------------------------------------------------------------
def _pot(D, n, b=None, X0=None):
    from mjhmc_amd.misc.distributions import ProductOfT
    W, lognu = ref_init_weights(D, D)

    class Fixed(ProductOfT):
        def init_X(self):
            self.Xinit = X0 if X0 is not None else np.zeros((D, n))

    return Fixed(ndims=D, nbasis=D, nbatch=n, lognu=lognu, W=W, b=b, state_dtype='float32'), W, lognu


@pytest.mark.parametrize('tag,D,n,src', [('pot_36x25', 36, 25, 'g2_dense'), ('pot_512x64', 512, 64, 'g2_dense'),
                                         ('pot36', 36, 256, 'ref_init_states')])
def test_pot_single_evaluation_matches_autograd_fixture(tag, D, n, src):
    g = load(src)
    X = g[tag + '_X']
    b = g[tag + '_b'] if tag + '_b' in g else None
    d, W, lognu = _pot(D, n, b)
    E, G = d.E(X), d.dEdX(X)
    assert E.shape == (1, n) and G.shape == (D, n)
    # float32 kernel against float64 autograd values: a few times the spread the reference's own float32 graph shows
    if tag + '_E32' in g:
        spread_E = max(rel(g[tag + '_E32'], g[tag + '_E']), 1e-7)
        spread_g = max(rel(g[tag + '_g32'], g[tag + '_g']), 1e-7)
    else:
        spread_E, spread_g = 1e-6, 4e-6
    assert rel(E[0], g[tag + '_E']) < 8 * spread_E, (rel(E[0], g[tag + '_E']), spread_E)
    assert rel(G, g[tag + '_g']) < 8 * spread_g, (rel(G, g[tag + '_g']), spread_g)


@pytest.mark.parametrize('D,N,state', [(768, 40, 'float64'), (1024, 70, 'float64'), (1024, 33, 'float32')])
def test_pot_more_than_512_dims(D, N, state):
    """distributions.py:379-406 takes any square size.  Beyond the 512 dims the register-resident tile kernels hold, the
    force is evaluated block by block (512 x 512 blocks of the pre-scaled matrices, the tile kernels' own GEMM:
    csrc/dense_pot.hip pot_big_eval) on the multi-pass path.  Single evaluations against the oracle's float32 force,
    sampling iterations against the oracle in the reference's arithmetic (float64 state around the float32 force);
    `state='float32'`: the same with the state rounded to float32 at the end of every iteration."""
    from mjhmc_amd.samplers import markov_jump_hmc as M
    from mjhmc_amd.misc.distributions import ProductOfT
    W, lognu = ref_init_weights(D, D)
    W = W + np.eye(D)
    rs = np.random.RandomState(D + N)
    X0 = rs.randn(D, N)
    b = 0.1 * rs.randn(D)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)           # noqa: E731

    class Fixed(ProductOfT):
        def init_X(self):
            self.Xinit = X0
    d = Fixed(ndims=D, nbasis=D, nbatch=N, lognu=lognu, W=W, b=b, state_dtype=state)
    en = orc.ProductOfT(W, lognu=lognu, b=b, force_dtype=np.float32)
    assert rel(d.E(X0)[0], en.E_val(X0)[0]) < 4e-6 and rel(d.dEdX(X0), en.dEdX_val(X0)) < 2e-5
    en64 = orc.ProductOfT(W, lognu=lognu, b=b, force_dtype=np.float64)            # and the float64 formulas, at float32 accuracy
    assert rel(d.E(X0)[0], en64.E_val(X0)[0]) < 2e-5 and rel(d.dEdX(X0), en64.dEdX_val(X0)) < 1e-4
    kw = dict(epsilon=0.1, beta=0.3, num_leapfrog_steps=5)
    s = M.MarkovJumpHMC(distribution=d, seed=17, resample=False, **kw)
    okw = dict(state_rounding=f32) if state == 'float32' else {}
    o = orc.MarkovJumpHMC(en, f32(X0) if state == 'float32' else X0, resample=False, rng=orc.PhiloxRNG(17, np.arange(N)), **dict(kw, **okw))
    _resync(s, o)
    ties = 0
    for t in range(4):
        ties += check_iteration(s, o, delta_rel=4e-6, x_tol=2e-6, e_rtol=4e-6, tag='pot %d %s it %d' % (D, state, t))
        assert s.l_count + s.f_count + s.r_count == (t + 1) * N
        _resync(s, o)
    assert ties <= 1
    X = s.state.X
    assert (np.abs(X - f32(X)).max() == 0) == (state == 'float32')
    Z = s.state.copy().L()                                     # HMCState.L() on a snapshot: the stand-alone leapfrog operator
    Zo = o.state.clone().L()
    assert rel(Z.X, Zo.X) < 2e-6 and rel(Z.EX, Zo.EX) < 4e-6
    out = s.sample(3, preserve_order=True)
    assert out.shape == (D, N, 3) and np.isfinite(out).all()


# ---------------------------------------------------------------------------------------------
# SparseImageCode: single evaluations
# ---------------------------------------------------------------------------------------------
def _sic(P, n, cauchy, X0, state_dtype='bfloat16'):
    from mjhmc_amd.misc.distributions import SparseImageCode
    B, imgs, a0 = sic_problem(0, n_patches=P)
    d = SparseImageCode(n_patches=P, n_batches=n, cauchy=cauchy, n_basis=1024, basis=B, imgs=imgs, init=X0, state_dtype=state_dtype)
    return d, B, imgs


def to_f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


# what storing the state does to it: SparseImageCode(state_dtype=...) -> the oracle's state_rounding
_SIC_STATES = {'bfloat16': to_bf16, 'float32': to_f32}


def _sic_cases():
    for P, n, cauchy in ((1, 1, True), (1, 8, True), (1, 8, False), (9, 1, True), (9, 4, True)):
        yield 'sic_p%d_n%d_%s' % (P, n, 'cauchy' if cauchy else 'laplace'), P, n, cauchy, 'g2_dense'
    yield 'sic_mj', 1, 10, True, 'ref_init_states'
    yield 'sic_ctl', 1, 10, True, 'ref_init_states'


@pytest.mark.parametrize('tag,P,n,cauchy,src', list(_sic_cases()), ids=[c[0] for c in _sic_cases()])
def test_sic_single_evaluation(tag, P, n, cauchy, src):
    g = load(src)
    X = g[tag + '_X']
    d, B, imgs = _sic(P, n, cauchy, X)
    E, G = d.E(X), d.dEdX(X)
    assert E.shape == (1, n) and G.shape == (P * 1024, n)
    # (1) mixed-precision restatement: bf16 operands, exact accumulation -> float32-accumulation tolerance
    o = orc.SparseImageCode(B, imgs[:, :P].T, lmbda=0.01, cauchy=cauchy, operand_rounding=to_bf16)
    Xb = to_bf16(X)                                   # what the device stores
    assert rel(E[0], o.E_val(Xb)[0]) < 2e-5, rel(E[0], o.E_val(Xb)[0])
    assert rel(G, o.dEdX_val(Xb)) < 2e-4, rel(G, o.dEdX_val(Xb))
    # (2) the full-precision autograd fixture at bf16 tolerance (8 significant bits in every operand)
    assert rel(E[0], g[tag + '_E']) < 4e-3, rel(E[0], g[tag + '_E'])
    assert rel(G, g[tag + '_g']) < 2e-2, rel(G, g[tag + '_g'])


@pytest.mark.parametrize('cls_name,D,N', [('MarkovJumpHMC', 36, 256), ('MarkovJumpHMC', 300, 70), ('ControlHMC', 36, 100),
                                          ('MarkovJumpHMC', 512, 96), ('ControlHMC', 512, 40)])
def test_pot_with_the_references_arithmetic_float64_state_float32_force(cls_name, D, N):
    """distributions.py:408-415 with hmc_state.py:29-38: the reference integrates float64 HMCState arrays around a float32
    Theano force.  ProductOfT(state_dtype='float64') does the same on the device (float32 matrix-core force on a
    downcast copy of the state, float64 kick / drift) and is compared with the oracle run WITHOUT any state rounding:
    the only difference left is the summation order inside the float32 matrix products."""
    from mjhmc_amd.samplers import markov_jump_hmc as M
    from mjhmc_amd.misc.distributions import ProductOfT
    W, lognu = ref_init_weights(D, D)
    W = W + np.eye(D)
    X0 = load('ref_init_states')['pot36_X'][:, :N] if D == 36 else np.random.RandomState(5).randn(D, N)

    class Fixed(ProductOfT):
        def init_X(self):
            self.Xinit = X0
    d = Fixed(ndims=D, nbasis=D, nbatch=N, lognu=lognu, W=W, state_dtype='float64')
    en = orc.ProductOfT(W, lognu=lognu, force_dtype=np.float32)
    assert rel(d.E(X0)[0], en.E_val(X0)[0]) < 4e-6 and rel(d.dEdX(X0), en.dEdX_val(X0)) < 2e-5
    kw = dict(epsilon=0.1, beta=0.3, num_leapfrog_steps=6)
    if cls_name == 'MarkovJumpHMC':
        s = M.MarkovJumpHMC(distribution=d, seed=17, resample=False, **kw)
        o = orc.MarkovJumpHMC(en, X0, resample=False, rng=orc.PhiloxRNG(17, np.arange(N)), **kw)
        assert np.allclose(s.state.V, o.state.V, rtol=0, atol=1e-13)   # float64 normals (device log / sincos within 1 ulp), no state rounding
        _resync(s, o)
        ties = 0
        for t in range(6):
            ties += check_iteration(s, o, delta_rel=4e-6, x_tol=1e-6, e_rtol=4e-6, tag='pot f64 state it %d' % t)
            _resync(s, o)
        assert ties <= 0.002 * 6 * N + 2, ties
        assert s.state.X.dtype == np.float64 and np.abs(s.state.X - s.state.X.astype(np.float32)).max() > 0   # not float32 values
    else:
        s = M.ControlHMC(distribution=d, seed=17, **kw)
        o = orc.ControlHMC(en, X0, rng=orc.PhiloxRNG(17, np.arange(N)), **kw)
        for t in range(6):
            check_control_iteration(s, o, delta_rel=4e-6, x_tol=1e-6, e_rtol=4e-6, tag='pot f64 control it %d' % t)
            _resync(s, o)
    out = s.sample(3, preserve_order=True) if cls_name == 'ControlHMC' else s.sample(3, preserve_order=True)
    assert out.shape == (D, N, 3) and np.isfinite(out).all()


@pytest.mark.parametrize('D,N,mode', [(36, 70, 'MJHMC'), (36, 20000, 'MJHMC'), (200, 300, 'MJHMC'), (512, 100, 'MJHMC'),
                                      (36, 100, 'CONTROL'), (512, 70, 'CONTROL'), (200, 65, 'CTHMC')])
def test_pot64_fused_equals_multipass(D, N, mode, monkeypatch):
    """ProductOfT with float64 state: the tile kernel with the state rows streamed through its epilogue
    (dense_pot64.hip -- what mjhmc_iterate runs) against the multi-pass form of the same arithmetic (host_energy.hip:
    one row pass + one force evaluation per leapfrog step; the test build's MJHMC_POT64_MULTIPASS=1).  Same operations
    in the same order around the same float32 GEMM code: X, V, dE/dX, E(X) and the transitions bit for bit (the kinetic
    energy's 512 squares are added up in a different order: 1e-12), every sampler family, ragged batches, the split
    schedule of big batches (N = 20000: two halves on two streams), one and several iterations per call, with and
    without a sample ring."""
    from mjhmc_amd import engine, _lib
    ctxs = (engine.context(0), hooks_context(0))
    W, lognu = ref_init_weights(D, D)
    W = W + np.eye(D)
    params = np.concatenate([[float(D)], W.ravel(), np.exp(lognu), 0.1 * np.random.RandomState(1).randn(D)])
    ens = [engine.DeviceEnergy(c, _lib.E_PRODUCT_OF_T, D, params) for c in ctxs]
    X0 = np.random.RandomState(3).randn(D, N)
    pair = [engine.DeviceSampler(en, X0, seed=8, dtype='float64', mode=getattr(_lib, 'MODE_' + mode)) for en in ens]
    fields = ('X', 'V', 'DEDX', 'EX', 'EV', 'HFLF', 'DWELL', 'TRANS')
    all_stats = [[], []]
    ring = N <= 300
    if ring:
        for s in pair:
            s.ring_alloc(4)
    for n_it, slot in ((1, -1), (3, 0), (2, -1), (1, 3)):
        if slot >= 0 and not ring:
            slot = -1
        for k, s in enumerate(pair):
            s.set_hparams(0.1, 5, 0.2, 0.3) if mode == 'CONTROL' else s.set_hparams(0.1, 5, 0.2, 1.0)
            if k == 1:
                monkeypatch.setenv('MJHMC_POT64_MULTIPASS', '1')
            else:
                monkeypatch.delenv('MJHMC_POT64_MULTIPASS', raising=False)
            st, done = s.iterate(n_it, ring_slot0=slot)
            assert done == n_it
            all_stats[k] += [(t.l, t.f, t.r, t.fl, t.n_cold, t.E_evals, t.dEdX_evals) for t in st]
        monkeypatch.delenv('MJHMC_POT64_MULTIPASS', raising=False)
        for f in fields:
            fa, fb = pair[0].read(getattr(_lib, 'F_' + f)), pair[1].read(getattr(_lib, 'F_' + f))
            if f in ('EV', 'HFLF', 'DWELL'):    # sum(V^2) is added up in a different order (tile reduction / one wavefront per row); an F clock's rate is a difference of two near-equal rates
                assert np.allclose(fa, fb, rtol=1e-7 if f == 'DWELL' else 1e-12, atol=0, equal_nan=True), (n_it, f)
            else:
                assert np.array_equal(fa, fb, equal_nan=True), (n_it, f, np.abs(fa - fb).max())
    assert all_stats[0] == all_stats[1]
    if ring:
        ra, rb = pair[0].ring_read(0, 4), pair[1].ring_read(0, 4)
        assert np.array_equal(ra, rb)
        assert np.array_equal(ra[:, -N:], pair[0].read(_lib.F_X))      # the last recorded sample is the live state
    X = pair[0].read(_lib.F_X)
    assert np.abs(X - X.astype(np.float32)).max() > 0         # float64 state, not float32 values
    for s in pair:
        s.close()


# ---------------------------------------------------------------------------------------------
# sampling iterations from the states the reference ships, transitions proven equal up to near ties
# ---------------------------------------------------------------------------------------------
def test_pot_iterations_from_the_reference_states():
    """MJHMC on ProductOfT 36 x 36 with the seed-2015 weights (search/MJHMC_poe_36/mjhmc_objective.py:15-23), started
    from the burn-in end points the reference ships (initializations/ProductOfT_...pickle[0])."""
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    X0 = load('ref_init_states')['pot36_X']
    N = X0.shape[1]
    d, W, lognu = _pot(36, N, None, X0)
    en = orc.ProductOfT(W, lognu=lognu, force_dtype=np.float64)
    s = MarkovJumpHMC(distribution=d, epsilon=0.1, beta=0.3, num_leapfrog_steps=6, seed=17, resample=False)
    o = orc.MarkovJumpHMC(en, X0, epsilon=0.1, beta=0.3, num_leapfrog_steps=6, resample=False,
                          rng=orc.PhiloxRNG(17, np.arange(N)), state_rounding=lambda a: a.astype(np.float32).astype(np.float64))
    assert np.allclose(s.state.V, o.state.V, atol=1e-6)
    _resync(s, o)
    ties = 0
    for t in range(6):
        ties += check_iteration(s, o, delta_rel=2e-5, x_tol=2e-5, e_rtol=2e-5, tag='pot36 it %d' % t)
        assert s.l_count + s.f_count + s.r_count == (t + 1) * N
        _resync(s, o)
    assert ties <= 0.002 * 6 * N + 2, ties            # near ties are rare by construction


@pytest.mark.parametrize('state', ['bfloat16', 'float32'])
def test_sic_iterations_from_the_reference_states(state):
    """MJHMC on SparseImageCode (one patch, 1024 coefficients) from the end points the reference ships
    (initializations/SparseImageCode_...pickle[0] and [3]; the dictionary itself is not in the reference checkout).
    `state`: the benchmark's bfloat16 state rows, or float32 rows as the reference's TensorFlow placeholders hold them
    (the matrix-core operands are bf16 either way)."""
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    to_state = _SIC_STATES[state]
    r = load('ref_init_states')
    X0 = to_state(np.concatenate([r['sic_mj_X'], r['sic_ctl_X']], axis=1))
    N = X0.shape[1]
    d, B, imgs = _sic(1, N, True, X0, state)
    en = orc.SparseImageCode(B, imgs[:, :1].T, lmbda=0.01, cauchy=True, operand_rounding=to_bf16)
    eps, L = 0.0625, 8             # a power of two: rounding the scaled residual == scaling the rounded residual
    s = MarkovJumpHMC(distribution=d, epsilon=eps, beta=0.2, num_leapfrog_steps=L, seed=23, resample=False)
    o = orc.MarkovJumpHMC(en, X0, epsilon=eps, beta=0.2, num_leapfrog_steps=L, resample=False,
                          rng=orc.PhiloxRNG(23, np.arange(N)), state_rounding=to_state)
    _resync(s, o)
    for t in range(5):
        check_iteration(s, o, delta_rel=5e-4, x_tol=1.0 / 128, e_rtol=5e-4, tag='sic %s it %d' % (state, t))
        assert s.l_count + s.f_count + s.r_count == (t + 1) * N
        _resync(s, o)


# ---------------------------------------------------------------------------------------------
# the comparison arms of the reference's ProductOfT and sparse-coding experiments (search/control_poe_36/
# mjhmc_objective.py:14, search/control_sp_img/control_objective.py:10) and the other sampler families on the
# matrix-core kernels
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cls_name,D,N', [('ControlHMC', 36, 70), ('HMC', 36, 40), ('HMCBase', 200, 33), ('ControlHMC', 512, 40)])
def test_pot_discrete_time_samplers_vs_oracle(cls_name, D, N):
    from mjhmc_amd.samplers import markov_jump_hmc as M
    X0 = np.random.RandomState(D + N).randn(D, N)
    d, W, lognu = _pot(D, N, None, X0)
    en = orc.ProductOfT(W, lognu=lognu, force_dtype=np.float64)
    kw = dict(epsilon=0.1, beta=0.6, num_leapfrog_steps=5)
    s = getattr(M, cls_name)(distribution=d, seed=41, **kw)
    o = getattr(orc, cls_name)(en, X0, rng=orc.PhiloxRNG(41, np.arange(N)),
                               state_rounding=lambda a: a.astype(np.float32).astype(np.float64), **kw)
    assert (s.beta, s.p_r, s.p_flip) == (o.beta, o.p_r, o.p_flip)
    _resync(s, o)
    ties = 0
    for t in range(6):
        before = (d.E_count, d.dEdX_count, en.E_count, en.dEdX_count)
        ties += check_control_iteration(s, o, delta_rel=2e-5, x_tol=2e-5, e_rtol=2e-5, tag='%s pot it %d' % (cls_name, t))
        if ties == 0:
            assert (s.l_count, s.f_count, s.r_count, s.fl_count) == (o.l_count, o.f_count, o.r_count, o.fl_count)
        assert (d.E_count - before[0], d.dEdX_count - before[1]) == (en.E_count - before[2], en.dEdX_count - before[3]) == (N, 5 * N)
        _resync(s, o)
    assert ties <= 1
    assert s.r_count > 0 or cls_name != 'ControlHMC'


def test_pot_continuous_time_sampler_vs_oracle():
    from mjhmc_amd.samplers.markov_jump_hmc import ContinuousTimeHMC
    D, N = 36, 60
    X0 = np.random.RandomState(5).randn(D, N)
    d, W, lognu = _pot(D, N, None, X0)
    en = orc.ProductOfT(W, lognu=lognu, force_dtype=np.float64)
    kw = dict(epsilon=0.1, beta=0.3, num_leapfrog_steps=5, resample=False)
    s = ContinuousTimeHMC(distribution=d, seed=43, **kw)
    o = orc.ContinuousTimeHMC(en, X0, rng=orc.PhiloxRNG(43, np.arange(N)),
                              state_rounding=lambda a: a.astype(np.float32).astype(np.float64), **kw)
    _resync(s, o)
    for t in range(5):
        s.sampling_iteration()
        o.sampling_iteration()
        # the F clock has rate 1 and R rate p_r: only the FL clock sees energies; with 60 particles a near tie is not expected
        assert (s.fl_count, s.f_count, s.r_count) == (o.fl_count, o.f_count, o.r_count), t
        assert np.abs(s.state.X - o.state.X).max() <= 2e-5 * max(1.0, np.abs(o.state.X).max())
        assert np.allclose(s.dwelling_times, o.dwelling_times, rtol=1e-3)
        _resync(s, o)


@pytest.mark.parametrize('cls_name,state', [('ControlHMC', 'bfloat16'), ('HMC', 'bfloat16'), ('ControlHMC', 'float32')])
def test_sic_discrete_time_samplers_vs_oracle(cls_name, state):
    from mjhmc_amd.samplers import markov_jump_hmc as M
    N = 40
    to_state = _SIC_STATES[state]
    B, imgs, a0 = sic_problem(0)
    X0 = to_state(a0[:, None] + 0.2 * np.random.RandomState(2).randn(1024, N))
    d, B, imgs = _sic(1, N, True, X0, state)
    en = orc.SparseImageCode(B, imgs[:, :1].T, lmbda=0.01, cauchy=True, operand_rounding=to_bf16)
    kw = dict(epsilon=0.0625, beta=0.6, num_leapfrog_steps=6)
    s = getattr(M, cls_name)(distribution=d, seed=47, **kw)
    o = getattr(orc, cls_name)(en, X0, rng=orc.PhiloxRNG(47, np.arange(N)), state_rounding=to_state, **kw)
    _resync(s, o)
    ties = 0
    for t in range(5):
        ties += check_control_iteration(s, o, delta_rel=5e-4, x_tol=1.0 / 128, e_rtol=5e-4, tag='%s sic %s it %d' % (cls_name, state, t))
        _resync(s, o)
    assert ties <= 1
    assert s.r_count > 0 or cls_name != 'ControlHMC'


# ---------------------------------------------------------------------------------------------
# HMCState operators on snapshots of dense-energy samplers (figures/poe_fig.py:58-76 assigns
# `sampler.state = HMCState(x, sampler)` on a ProductOfT sampler and integrates it)
# ---------------------------------------------------------------------------------------------
def test_pot_state_assignment_and_leapfrog_operator():
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.samplers.hmc_state import HMCState
    D, N = 36, 45
    X0 = np.random.RandomState(6).randn(D, N)
    d, W, lognu = _pot(D, N, None, X0)
    en = orc.ProductOfT(W, lognu=lognu, force_dtype=np.float64)
    s = MarkovJumpHMC(distribution=d, epsilon=0.1, beta=0.3, num_leapfrog_steps=7, seed=3, resample=False)
    Xn = np.random.RandomState(7).randn(D, N) * 1.3
    Vn = np.random.RandomState(8).randn(D, N)
    s.state = HMCState(Xn, s, V=Vn)                       # poe_fig.py:59
    f32 = lambda a: a.astype(np.float32).astype(np.float64)   # noqa: E731
    assert np.array_equal(s.state.X, f32(Xn)) and np.array_equal(s.state.V, f32(Vn))
    o = orc.MarkovJumpHMC(en, f32(Xn), epsilon=0.1, beta=0.3, num_leapfrog_steps=7, V0=f32(Vn), resample=False,
                          rng=orc.ReplayRNG())
    assert rel(s.state.EX, o.state.EX) < 2e-5 and rel(s.state.dEdX, o.state.dEdX) < 2e-5
    Z = s.state.copy().L()
    Zo = o.state.clone().L()
    assert rel(Z.X, Zo.X) < 2e-5 and rel(Z.V, Zo.V) < 2e-5
    assert rel(Z.EX, Zo.EX) < 2e-5 and rel(Z.EV, Zo.EV) < 2e-5 and rel(Z.dEdX, Zo.dEdX) < 5e-5
    Z1 = s.state.copy()
    Z1.leapfrog()                                          # a single step; energies are not refreshed (hmc_state.py:86-91)
    o1 = o.state.clone()
    o1.leap()
    assert rel(Z1.X, o1.X) < 2e-5 and rel(Z1.V, o1.V) < 2e-5


def test_sic_leapfrog_operator():
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    N = 20
    B, imgs, a0 = sic_problem(0)
    X0 = to_bf16(a0[:, None] + 0.2 * np.random.RandomState(2).randn(1024, N))
    d, B, imgs = _sic(1, N, True, X0)
    en = orc.SparseImageCode(B, imgs[:, :1].T, lmbda=0.01, cauchy=True, operand_rounding=to_bf16)
    s = MarkovJumpHMC(distribution=d, epsilon=0.0625, beta=0.3, num_leapfrog_steps=6, seed=3, resample=False)
    o = orc.MarkovJumpHMC(en, X0, epsilon=0.0625, beta=0.3, num_leapfrog_steps=6, V0=s.state.V, resample=False,
                          rng=orc.ReplayRNG(), state_rounding=to_bf16)
    Z = s.state.copy().L()
    Zo = o.state.clone().L()
    assert np.abs(Z.X - Zo.X).max() <= np.abs(Zo.X).max() / 128 and np.abs(Z.V - Zo.V).max() <= np.abs(Zo.V).max() / 128
    scale = float(np.abs(Zo.H()).max())
    assert np.abs(Z.EX - Zo.EX).max() <= 5e-4 * scale and np.abs(Z.EV - Zo.EV).max() <= 5e-4 * scale


# ---------------------------------------------------------------------------------------------
# SparseImageCode with the reference's default of nine patches per particle (tf_distributions.py:208,
# experiments/spectral.py:247): 27 of a tile's 32 columns work, three particles per tile
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('P,N', [(9, 7), (2, 33), (3, 10)])
def test_sic_several_patches_iterations_vs_oracle(P, N):
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    B, imgs, a0 = sic_problem(0, n_patches=P)
    X0 = to_bf16(a0[:, None] + 0.2 * np.random.RandomState(P).randn(P * 1024, N))
    d, B, imgs = _sic(P, N, True, X0)
    en = orc.SparseImageCode(B, imgs[:, :P].T, lmbda=0.01, cauchy=True, operand_rounding=to_bf16)
    kw = dict(epsilon=0.0625, beta=0.2, num_leapfrog_steps=6, resample=False)
    s = MarkovJumpHMC(distribution=d, seed=51, **kw)
    o = orc.MarkovJumpHMC(en, X0, rng=orc.PhiloxRNG(51, np.arange(N)), state_rounding=to_bf16, **kw)
    assert np.abs(s.state.V - o.state.V).max() < 3e-2                # bf16-rounded tick-0 momentum, all P * 1024 dims
    _resync(s, o)
    assert np.allclose(s.state.EX, o.state.EX, rtol=1e-4) and np.allclose(s.state.EV, o.state.EV, rtol=1e-5)
    for t in range(4):
        check_iteration(s, o, delta_rel=5e-4, x_tol=1.0 / 128, e_rtol=5e-4, tag='sic P=%d it %d' % (P, t))
        assert s.l_count + s.f_count + s.r_count == (t + 1) * N
        _resync(s, o)
    out = s.sample(3)                                                # ring slots of P * 1024-wide rows
    assert out.shape == (P * 1024, 3 * N) and np.array_equal(out[:, -N:], s.state.X)


# ---------------------------------------------------------------------------------------------
# the 512-atom dictionary (the reference accepts n_basis in [1024, 512], tf_distributions.py:219)
# ---------------------------------------------------------------------------------------------
def _sic512(P, n, cauchy, X0):
    from mjhmc_amd.misc.distributions import SparseImageCode
    B, imgs, a0 = sic_problem(3, n_patches=P, n_coeffs=512)
    return SparseImageCode(n_patches=P, n_batches=n, cauchy=cauchy, n_basis=512, basis=B, imgs=imgs, init=X0, state_dtype='bfloat16'), B, imgs, a0


@pytest.mark.parametrize('P,n,cauchy', [(1, 40, True), (1, 5, False), (9, 4, True)])
def test_sic_512_atoms_single_evaluation(P, n, cauchy):
    from oracle import autograd_energies as ag
    B, imgs, a0 = sic_problem(3, n_patches=P, n_coeffs=512)
    X = a0[:, None] + 0.3 * np.random.RandomState(8).randn(P * 512, n)
    d, B, imgs, _ = _sic512(P, n, cauchy, X)
    E, G = d.E(X), d.dEdX(X)
    assert E.shape == (1, n) and G.shape == (P * 512, n)
    o = orc.SparseImageCode(B, imgs[:, :P].T, lmbda=0.01, cauchy=cauchy, operand_rounding=to_bf16)
    Xb = to_bf16(X)
    assert rel(E[0], o.E_val(Xb)[0]) < 2e-5 and rel(G, o.dEdX_val(Xb)) < 2e-4
    # the reference's forward graph (per column) under autograd, at bf16 tolerance
    Ea, ga = ag.sparse_image_code_per_column(B, imgs[:, :P].T, 0.01, cauchy, X)
    assert rel(E[0], Ea) < 4e-3 and rel(G, ga) < 2e-2


@pytest.mark.parametrize('P,N', [(1, 70), (9, 7)])
def test_sic_512_atoms_iterations_vs_oracle(P, N):
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    B, imgs, a0 = sic_problem(3, n_patches=P, n_coeffs=512)
    X0 = to_bf16(a0[:, None] + 0.2 * np.random.RandomState(P).randn(P * 512, N))
    d, B, imgs, _ = _sic512(P, N, True, X0)
    en = orc.SparseImageCode(B, imgs[:, :P].T, lmbda=0.01, cauchy=True, operand_rounding=to_bf16)
    kw = dict(epsilon=0.0625, beta=0.2, num_leapfrog_steps=6, resample=False)
    s = MarkovJumpHMC(distribution=d, seed=52, **kw)
    o = orc.MarkovJumpHMC(en, X0, rng=orc.PhiloxRNG(52, np.arange(N)), state_rounding=to_bf16, **kw)
    assert np.abs(s.state.V - o.state.V).max() < 3e-2
    _resync(s, o)
    for t in range(4):
        check_iteration(s, o, delta_rel=5e-4, x_tol=1.0 / 128, e_rtol=5e-4, tag='sic512 P=%d it %d' % (P, t))
        assert s.l_count + s.f_count + s.r_count == (t + 1) * N
        _resync(s, o)
    out = s.sample(3)
    assert out.shape == (P * 512, 3 * N) and np.array_equal(out[:, -N:], s.state.X)


def test_sic_512_atoms_control_arm_and_leapfrog():
    from mjhmc_amd.samplers.markov_jump_hmc import ControlHMC
    N = 40
    B, imgs, a0 = sic_problem(3, n_coeffs=512)
    X0 = to_bf16(a0[:, None] + 0.2 * np.random.RandomState(2).randn(512, N))
    d, B, imgs, _ = _sic512(1, N, True, X0)
    en = orc.SparseImageCode(B, imgs[:, :1].T, lmbda=0.01, cauchy=True, operand_rounding=to_bf16)
    kw = dict(epsilon=0.0625, beta=0.3, num_leapfrog_steps=5)
    s = ControlHMC(distribution=d, seed=9, **kw)
    o = orc.ControlHMC(en, X0, rng=orc.PhiloxRNG(9, np.arange(N)), state_rounding=to_bf16, **kw)
    _resync(s, o)
    for t in range(3):
        check_control_iteration(s, o, delta_rel=5e-4, x_tol=1.0 / 128, e_rtol=5e-4, tag='sic512 control it %d' % t)
        _resync(s, o)
    # HMCState.L() on a snapshot
    Z = s.state.copy().L()
    Zo = o.state.clone().L()
    assert np.abs(Z.X - Zo.X).max() <= np.abs(Zo.X).max() / 128
    scale = float(np.abs(Zo.H()).max())
    assert np.abs(Z.EX - Zo.EX).max() <= 5e-4 * scale and np.abs(Z.EV - Zo.EV).max() <= 5e-4 * scale


# ---------------------------------------------------------------------------------------------
# dense batches are launched as free-running parts on their own streams (api.hip: part_args): invisible in the results
# ---------------------------------------------------------------------------------------------
def _dense_case(what, ctxs, seed=8):
    """(energies per context, X0, dtype, mode, (eps, L, beta)) of the split / shortcut tests"""
    from mjhmc_amd import engine, _lib
    mode = {'pot36_control': _lib.MODE_CONTROL, 'sic_p1_ct': _lib.MODE_CTHMC}.get(what, _lib.MODE_MJHMC)   # the other sampler families
    what = what.split('_c')[0]
    if what in ('pot36', 'pot36f64'):
        D, N, dtype = 36, 20000, 'float64' if what == 'pot36f64' else 'float32'
        W, lognu = ref_init_weights(D, D)
        params = np.concatenate([[float(D)], W.ravel(), np.exp(lognu), np.zeros(D)])
        ens = [engine.DeviceEnergy(c, _lib.E_PRODUCT_OF_T, D, params) for c in ctxs]
        X0 = np.random.RandomState(3).randn(D, N)
        hp = (0.1, 6, 0.1)
    else:
        P, N = (1, 17000) if what == 'sic_p1' else (9, 6100)
        B, imgs, a0 = sic_problem(0, n_patches=P)
        D, dtype = P * 1024, 'bfloat16'
        params = np.concatenate([[float(P), 256.0, 1024.0, 0.01, 1.0], B.ravel(), imgs[:, :P].T.ravel()])
        ens = [engine.DeviceEnergy(c, _lib.E_SPARSE_CODE, D, params) for c in ctxs]
        X0 = a0[:, None] + 0.2 * np.random.RandomState(4).randn(D, N)
        hp = (0.0625, 3, 0.1)
    return ens, X0, dtype, mode, hp


_FIELDS = ('X', 'V', 'EX', 'EV', 'HFLF', 'CACHE', 'DWELL', 'TRANS')


@pytest.mark.parametrize('what,parts', [('pot36', None), ('pot36f64', None), ('sic_p1', None), ('sic_p9', None), ('pot36_control', None),
                                        ('pot36f64_control', None), ('sic_p1_ct', None),
                                        ('pot36', 3), ('pot36f64', 4), ('sic_p1', 4), ('sic_p9', 3), ('pot36f64_control', 3)])
def test_split_launches_equal_single_launches(what, parts, monkeypatch):
    """[0] the product library's own schedule (two parts at these sizes) -- or, parts = 3 / 4, the test build told to run
    that many -- against [1] the test build with MJHMC_NO_SPLIT: one launch sequence on one stream."""
    from mjhmc_amd import engine, _lib
    ctxs = (engine.context(0) if parts is None else hooks_context(0), hooks_context(0))
    ens, X0, dtype, mode, hp = _dense_case(what, ctxs)
    pair = [engine.DeviceSampler(en, X0, seed=8, dtype=dtype, mode=mode) for en in ens]
    all_stats = [[], []]
    for n_it in (1, 3, 2):
        for k, s in enumerate(pair):
            s.set_hparams(hp[0], hp[1], hp[2], 1.0)
            monkeypatch.delenv('MJHMC_NO_SPLIT', raising=False)
            monkeypatch.delenv('MJHMC_SPLIT_PARTS', raising=False)
            if k == 1:
                monkeypatch.setenv('MJHMC_NO_SPLIT', '1')
            elif parts:
                monkeypatch.setenv('MJHMC_SPLIT_PARTS', str(parts))
            st, done = s.iterate(n_it)
            assert done == n_it
            all_stats[k] += [(t.l, t.f, t.r, t.n_cold, t.E_evals, t.dEdX_evals, t.n_flf_run) for t in st]
        monkeypatch.delenv('MJHMC_NO_SPLIT', raising=False)
        monkeypatch.delenv('MJHMC_SPLIT_PARTS', raising=False)
        for f in _FIELDS:
            fa, fb = pair[0].read(getattr(_lib, 'F_' + f)), pair[1].read(getattr(_lib, 'F_' + f))
            assert np.array_equal(fa, fb, equal_nan=True), (what, n_it, f)
    assert all_stats[0] == all_stats[1]
    for s in pair:
        s.close()


@pytest.mark.parametrize('what', ['pot36', 'pot36f64', 'sic_p1', 'sic_p9'])
def test_f_mover_shortcut_is_bit_identical(what, monkeypatch):
    """A particle that has just moved by F needs no inverse-L trajectory: F L F (X, -V) = F L (X, V) is the L proposal of
    the iteration in which it flipped, and the dense kernels hand its H() on (dense_pot.hip).  [0] the product library
    against [1] the test build with MJHMC_NO_FSPEC=1, which integrates every cold particle's inverse-L proposal as the
    reference does (hmc_state.py:109-119): state, cache, dwelling times, transitions and the reference's counters bit
    for bit -- through single iterations, batches, a change of the step size (which drops the hand-over), a
    reset_flf_cache and a checkpoint / restore; only n_flf_run differs: it is n_cold on the one side, the R-movers of
    the iteration before on the other."""
    from mjhmc_amd import engine, _lib
    ctxs = (engine.context(0), hooks_context(0))
    ens, X0, dtype, mode, hp = _dense_case(what, ctxs)
    pair = [engine.DeviceSampler(en, X0, seed=8, dtype=dtype, mode=mode) for en in ens]
    N = X0.shape[1]
    stats = [[], []]
    saved = 0.0

    def step(n_it, eps=hp[0], L=hp[1]):
        for k, s in enumerate(pair):
            s.set_hparams(eps, L, hp[2], 1.0)
            if k == 1:
                monkeypatch.setenv('MJHMC_NO_FSPEC', '1')
            else:
                monkeypatch.delenv('MJHMC_NO_FSPEC', raising=False)
            st, done = s.iterate(n_it)
            assert done == n_it
            stats[k] += list(st)
        monkeypatch.delenv('MJHMC_NO_FSPEC', raising=False)
        for f in _FIELDS:
            fa, fb = pair[0].read(getattr(_lib, 'F_' + f)), pair[1].read(getattr(_lib, 'F_' + f))
            assert np.array_equal(fa, fb, equal_nan=True), (what, len(stats[0]), f)

    step(1)
    step(4)
    step(1)
    n_a = len(stats[0])
    step(2, eps=hp[0] / 2, L=2 * hp[1])        # new step size: the hand-over is dropped, everything cold integrates
    assert stats[0][n_a].n_flf_run == stats[0][n_a].n_cold
    step(1)                                     # and back (the retry sequence of markov_jump_hmc.py:376-389)
    assert stats[0][-1].n_flf_run == stats[0][-1].n_cold
    for s in pair:
        s.checkpoint()
    step(3)
    for s in pair:
        s.restore()
    n_a = len(stats[0])
    step(2)
    assert stats[0][n_a].n_flf_run == stats[0][n_a].n_cold
    for s in pair:
        s.reset_flf_cache()
    step(2)
    assert stats[0][-2].n_cold == N and stats[0][-2].n_flf_run == N
    for a, b in zip(*stats):
        assert (a.l, a.f, a.r, a.n_cold, a.E_evals, a.dEdX_evals) == (b.l, b.f, b.r, b.n_cold, b.E_evals, b.dEdX_evals)
        assert b.n_flf_run == b.n_cold and a.n_flf_run <= a.n_cold
    # inside a batch and from one single-iteration call to the next: only the R-movers of the iteration before integrate
    a = stats[0]
    assert a[1].n_flf_run == a[0].r and a[1].n_cold == a[0].f + a[0].r
    assert a[2].n_flf_run == a[1].r and a[3].n_flf_run == a[2].r and a[5].n_flf_run == a[4].r
    saved = sum(t.n_cold - t.n_flf_run for t in a)
    assert saved > 0, 'no F move in the whole run: the test does not exercise the shortcut'
    for s in pair:
        s.close()


@pytest.mark.parametrize('what,poison', [('sic_p1', '4:300'), ('sic_p1', '6:16000'), ('pot36', '3:19000'), ('pot36', '0:5'),
                                         ('pot36f64', '5:19000'), ('pot36f64', '2:7')])
def test_split_launches_failure_in_the_middle_of_a_call(what, poison, monkeypatch):
    """A non-finite rate while the two halves of a big dense batch run freely: the call must end exactly like a call
    that was never split -- same `done`, same state, same tallies of the good iterations -- (api.hip: the state copy
    taken at the start of the call is put back and the call re-run on one stream).  The failure is placed with the
    library's test hook MJHMC_DEBUG_POISON=iteration:particle (that particle's kinetic energy reads NaN there)."""
    from mjhmc_amd import engine, _lib
    ctx = hooks_context(0)                 # both samplers from the test build: only it can place a failure
    if what in ('pot36', 'pot36f64'):
        D, N, dtype = 36, 20000, 'float64' if what == 'pot36f64' else 'float32'
        W, lognu = ref_init_weights(D, D)
        en = engine.DeviceEnergy(ctx, _lib.E_PRODUCT_OF_T, D, np.concatenate([[float(D)], W.ravel(), np.exp(lognu), np.zeros(D)]))
        X0 = np.random.RandomState(3).randn(D, N)
        hp = (0.1, 6, 0.1)
    else:
        N, dtype = 17000, 'bfloat16'
        B, imgs, a0 = sic_problem(0)
        en = engine.DeviceEnergy(ctx, _lib.E_SPARSE_CODE, 1024,
                                 np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), imgs[:, :1].T.ravel()]))
        X0 = a0[:, None] + 0.2 * np.random.RandomState(4).randn(1024, N)
        hp = (0.0625, 3, 0.1)
    it = int(poison.split(':')[0])
    pair = [engine.DeviceSampler(en, X0, seed=8, dtype=dtype) for _ in range(2)]
    res = []
    monkeypatch.setenv('MJHMC_DEBUG_POISON', poison)
    for k, s in enumerate(pair):
        s.set_hparams(hp[0], hp[1], hp[2], 1.0)
        if k == 1:
            monkeypatch.setenv('MJHMC_NO_SPLIT', '1')
        else:
            monkeypatch.delenv('MJHMC_NO_SPLIT', raising=False)
        st, done = s.iterate(8)
        res.append((done, [(t.l, t.f, t.r, t.n_cold, t.E_evals, t.dEdX_evals) for t in st[:done]], st[done].nonfinite))
    monkeypatch.delenv('MJHMC_NO_SPLIT', raising=False)
    monkeypatch.delenv('MJHMC_DEBUG_POISON', raising=False)
    assert res[0] == res[1] and res[0][0] == it and res[0][2] == 1
    fields = ('X', 'V', 'EX', 'EV', 'HFLF')
    for f in fields:
        fa, fb = pair[0].read(getattr(_lib, 'F_' + f)), pair[1].read(getattr(_lib, 'F_' + f))
        assert np.array_equal(fa, fb, equal_nan=True), f
    # the poisoned value is part of the committed state (it was an input of the failing iteration): heal it, then the
    # retry protocol continues identically in both forms
    for s in pair:
        s.write(_lib.F_V, s.read(_lib.F_V))            # re-derives EV
        s.set_hparams(hp[0] / 2, 2 * hp[1], hp[2], 1.0)
        s.reset_flf_cache()
    a, da = pair[0].iterate(3)
    monkeypatch.setenv('MJHMC_NO_SPLIT', '1')
    b, db = pair[1].iterate(3)
    monkeypatch.delenv('MJHMC_NO_SPLIT', raising=False)
    assert da == db == 3
    for f in fields + ('DWELL', 'TRANS'):
        assert np.array_equal(pair[0].read(getattr(_lib, 'F_' + f)), pair[1].read(getattr(_lib, 'F_' + f)), equal_nan=True), f
    for s in pair:
        s.close()


@pytest.mark.parametrize('what,poison', [('pot36', '3:19000'), ('pot36', '5:7'), ('pot36f64', '4:19000'), ('sic_p1', '4:300')])
def test_streamed_download_when_a_split_call_fails_in_the_middle(what, poison, monkeypatch):
    """mjhmc_iterate_download while the two halves of a dense batch run freely and one of them meets a non-finite rate:
    the call is put back and re-run on one stream (api.hip: iterate_t), and NOTHING the first run handed to the download
    thread may survive -- a lagging half left its columns of the slots stale (dl_restart).  The host array of the
    streamed call must hold, for the iterations that completed, exactly what a never-split call records in its ring."""
    from mjhmc_amd import engine, _lib
    ctx = hooks_context(0)
    if what in ('pot36', 'pot36f64'):
        D, N, dtype = 36, 20000, 'float64' if what == 'pot36f64' else 'float32'
        W, lognu = ref_init_weights(D, D)
        en = engine.DeviceEnergy(ctx, _lib.E_PRODUCT_OF_T, D, np.concatenate([[float(D)], W.ravel(), np.exp(lognu), np.zeros(D)]))
        X0 = np.random.RandomState(3).randn(D, N)
        hp = (0.1, 6, 0.1)
    else:
        D, N, dtype = 1024, 17000, 'bfloat16'
        B, imgs, a0 = sic_problem(0)
        en = engine.DeviceEnergy(ctx, _lib.E_SPARSE_CODE, 1024,
                                 np.concatenate([[1.0, 256.0, 1024.0, 0.01, 1.0], B.ravel(), imgs[:, :1].T.ravel()]))
        X0 = a0[:, None] + 0.2 * np.random.RandomState(4).randn(1024, N)
        hp = (0.0625, 3, 0.1)
    it, n = int(poison.split(':')[0]), 8
    pair = [engine.DeviceSampler(en, X0, seed=8, dtype=dtype) for _ in range(2)]
    for s in pair:
        s.set_hparams(hp[0], hp[1], hp[2], 1.0)
        s.ring_alloc(n)
    monkeypatch.setenv('MJHMC_DEBUG_POISON', poison)
    monkeypatch.delenv('MJHMC_NO_SPLIT', raising=False)
    out = np.full((D, n * N), -7.0)
    st_a, done_a = pair[0].iterate_download(n, 0, out)
    monkeypatch.setenv('MJHMC_NO_SPLIT', '1')
    st_b, done_b = pair[1].iterate(n, ring_slot0=0)
    monkeypatch.delenv('MJHMC_NO_SPLIT', raising=False)
    monkeypatch.delenv('MJHMC_DEBUG_POISON', raising=False)
    assert done_a == done_b == it
    assert [(t.l, t.f, t.r, t.n_cold) for t in st_a[:it]] == [(t.l, t.f, t.r, t.n_cold) for t in st_b[:it]]
    want = pair[1].ring_read(0, it)
    got = out[:, :it * N]
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError(('streamed slots differ from the recorded ring', len(bad), 'time slots', sorted(set((bad[:, 1] // N).tolist())),
                              'particles', sorted(set((bad[:, 1] % N).tolist()))[:8]))
    assert np.array_equal(pair[0].ring_read(0, it), want)          # and the device ring of the streamed call itself
    for f in ('X', 'V', 'EX', 'EV', 'HFLF'):
        assert np.array_equal(pair[0].read(getattr(_lib, 'F_' + f)), pair[1].read(getattr(_lib, 'F_' + f)), equal_nan=True), f
    for s in pair:
        s.close()


# ---------------------------------------------------------------------------------------------
# the REPLAY instances of the tile kernels (recorded random numbers fed from the host, the mode the golden replays of
# the elementwise energies run in): every draw is the caller's, so device and oracle can be compared decision by decision
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('what,cls_name', [('pot32', 'MarkovJumpHMC'), ('pot64', 'MarkovJumpHMC'), ('sic', 'MarkovJumpHMC'),
                                           ('pot32', 'ControlHMC'), ('pot64', 'ControlHMC'), ('sic', 'ControlHMC'),
                                           ('pot64', 'ContinuousTimeHMC'), ('pot32', 'ContinuousTimeHMC'),
                                           ('sic', 'ContinuousTimeHMC')])
def test_dense_kernels_in_replay_mode(what, cls_name):
    from mjhmc_amd.samplers import markov_jump_hmc as M
    rs = np.random.RandomState(123)
    T = 4
    if what == 'sic':
        N, D = 40, 1024
        B, imgs, a0 = sic_problem(0)
        X0 = to_bf16(a0[:, None] + 0.2 * rs.randn(D, N))
        d, B, imgs = _sic(1, N, True, X0)
        en = orc.SparseImageCode(B, imgs[:, :1].T, lmbda=0.01, cauchy=True, operand_rounding=to_bf16)
        rounding, tol = to_bf16, dict(delta_rel=5e-4, x_tol=1.0 / 128, e_rtol=5e-4)
        kw = dict(epsilon=0.0625, beta=0.3, num_leapfrog_steps=5)
        V0 = to_bf16(rs.randn(D, N))
        normals = [to_bf16(rs.randn(D, N)) for _ in range(T)]          # (what the device stores of them)
    else:
        N, D = 70, 36
        X0 = rs.randn(D, N)
        state = 'float64' if what == 'pot64' else 'float32'
        W, lognu = ref_init_weights(D, D)
        W = W + np.eye(D)
        from mjhmc_amd.misc.distributions import ProductOfT

        class Fixed(ProductOfT):
            def init_X(self):
                self.Xinit = X0
        d = Fixed(ndims=D, nbasis=D, nbatch=N, lognu=lognu, W=W, state_dtype=state)
        f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)       # noqa: E731
        if what == 'pot64':
            en, rounding, tol = orc.ProductOfT(W, lognu=lognu, force_dtype=np.float32), None, dict(delta_rel=4e-6, x_tol=1e-6, e_rtol=4e-6)
        else:
            en, rounding, tol = orc.ProductOfT(W, lognu=lognu, force_dtype=np.float64), f32, dict(delta_rel=2e-5, x_tol=2e-5, e_rtol=2e-5)
        kw = dict(epsilon=0.1, beta=0.3, num_leapfrog_steps=5)
        V0 = rs.randn(D, N) if rounding is None else rounding(rs.randn(D, N))
        normals = [rs.randn(D, N) if rounding is None else rounding(rs.randn(D, N)) for _ in range(T)]
    okw = {} if rounding is None else dict(state_rounding=rounding)
    if cls_name == 'ControlHMC':
        u_acc, u_flip, u_r = [rs.rand(N) for _ in range(T)], [rs.rand(N) for _ in range(T)], [rs.rand() for _ in range(T)]
        s = M.ControlHMC(distribution=d, Vinit=V0, seed=1, **kw)
        fired = [u < s.p_r for u in u_r]
        o = orc.ControlHMC(en, X0, V0=V0, rng=orc.ReplayRNG(normals=[normals[t] for t in range(T) if fired[t]],
                                                            uniforms=[v for t in range(T) for v in (u_acc[t], u_flip[t], u_r[t])]),
                           **dict(kw, **okw))
        _resync(s, o)
        for t in range(T):
            # the device side of check_control_iteration, with the recorded numbers
            H0, HL = o.state.H()[0].copy(), None
            s.sampling_iteration(replay=[(normals[t] if fired[t] else np.zeros((D, N)), np.concatenate([u_acc[t], u_flip[t], [u_r[t]]]))])
            o.sampling_iteration()
            tr = s._dev.read(8)
            acc_o, flip_o = np.isin(np.arange(N), o.last_fl_idx), np.isin(np.arange(N), o.last_flip_idx)
            assert np.array_equal(((tr >> 1) & 1).astype(bool), flip_o), t
            same = (tr & 1).astype(bool) == acc_o
            assert (~same).sum() <= 1, t                          # (an accept decision within the kernel's energy error of a tie)
            xs = max(1.0, float(np.abs(o.state.X).max()))
            assert np.abs(s.state.X[:, same] - o.state.X[:, same]).max() <= tol['x_tol'] * xs, t
            assert np.abs(s.state.V[:, same] - o.state.V[:, same]).max() <= tol['x_tol'] * max(1.0, float(np.abs(o.state.V).max())), t
            _resync(s, o)
        assert sum(fired) == 0 or s.r_count > 0
        return
    exps = [rs.standard_exponential((3, N)) for _ in range(T)]
    if cls_name == 'MarkovJumpHMC':
        s = M.MarkovJumpHMC(distribution=d, Vinit=V0, seed=1, resample=False, **kw)
        o = orc.MarkovJumpHMC(en, X0, V0=V0, resample=False, rng=orc.ReplayRNG(normals=normals, exps=exps), **dict(kw, **okw))
        _resync(s, o)
        ties = 0
        for t in range(T):
            # check_iteration drives both sides itself: hand the recorded numbers to the device through a one-shot patch
            step = s.sampling_iteration
            s.sampling_iteration = lambda step=step, t=t: step(replay=[(normals[min(o.rng.n_normals_used, T - 1)], exps[t])])
            try:
                ties += check_iteration(s, o, tag='%s replay it %d' % (what, t), **tol)
            finally:
                del s.sampling_iteration
            _resync(s, o)
        assert ties <= 1
    else:
        s = M.ContinuousTimeHMC(distribution=d, Vinit=V0, seed=1, resample=False, **kw)
        o = orc.ContinuousTimeHMC(en, X0, V0=V0, resample=False, rng=orc.ReplayRNG(normals=normals, exps=exps), **dict(kw, **okw))
        _resync(s, o)
        for t in range(T):
            s.sampling_iteration(replay=[(normals[min(o.rng.n_normals_used, T - 1)], exps[t])])
            o.sampling_iteration()
            same = s._dev.read(8) == np.array([1, 0, 2], dtype=np.uint8)[o.last_transition] if hasattr(o, 'last_transition') else None
            assert (s.fl_count, s.f_count, s.r_count) == (o.fl_count, o.f_count, o.r_count), t
            assert np.abs(s.state.X - o.state.X).max() <= tol['x_tol'] * max(1.0, np.abs(o.state.X).max()), t
            assert np.abs(s.state.V - o.state.V).max() <= tol['x_tol'] * max(1.0, np.abs(o.state.V).max()), t
            _resync(s, o)
