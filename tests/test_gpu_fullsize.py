"""GPU: BASELINE.json's full sizes (configs 2-4; configs[1] is in test_gpu_parity.py).  The oracle cannot run
10^5 .. 10^6 particles in seconds, so each test checks size-independent properties on the whole batch (counter
identities, cache bookkeeping, stored energies == energies re-evaluated from the stored state) and a random
column subset against the oracle driven by the same Philox streams (the RNG is keyed by global particle id).

Tolerances: float64 elementwise energies 1e-10 relative and bit-exact transitions; the dense energies run in
float32 / bf16 on the device and are compared with the float64 oracle from identical inputs as in
test_gpu_parity.py (transitions equal except at near ties)."""
import numpy as np
import pytest

import bench
from oracle import mjhmc_oracle as orc
from tests.test_gpu_parity import close, to_bf16
from tests.helpers import check_iteration, resync

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('calls', ['one iteration per call', 'one call'])
def test_full_size_c4_funnel(calls):
    """configs[3]: Neal funnel, ndims=32, nparticles=1000000, L=15, float64 (whole batch on one GPU) -- as single-iteration
    calls (a trajectory launch in row form + a jump-process launch each) and as ONE call (all iterations fused in one launch
    of mjhmc_fused_rows_kernel, what bench.py times): the same checks, the same oracle."""
    from mjhmc_amd import engine, _lib
    w = bench.WORKLOADS['c4']
    D, N, L, eps, beta = w['D'], w['N'], w['L'], w['eps'], w['beta']
    X0 = bench.initial_state(w, N, 0)
    ctx = engine.context(0)
    en = engine.DeviceEnergy(ctx, _lib.E_FUNNEL_NEAL, D, w['params'])
    s = engine.DeviceSampler(en, X0, seed=11)
    p_r = -np.log(1 - beta) * 0.5
    s.set_hparams(eps, L, p_r, 1.0)
    T = 3
    stats = []
    if calls == 'one call':
        stats, done = s.iterate(T)
        assert done == T
        stats = list(stats[:T])
    else:
        for _ in range(T):                   # single launches: the list's walkers integrate the cold caches' inverse-L proposals
            st, done = s.iterate(1)
            assert done == 1
            stats += st
    n_cold_expected = N
    for st in stats:
        assert st.l + st.f + st.r == N and st.nonfinite == 0
        assert st.n_cold == n_cold_expected
        assert st.E_evals == N + st.n_cold and st.dEdX_evals == L * (N + st.n_cold)
        n_cold_expected = N - st.l
    trans, cache = s.read(_lib.F_TRANS), s.read(_lib.F_CACHE)
    assert np.array_equal(cache == 1, trans == 0)
    X, V, EX, EV = s.read(_lib.F_X), s.read(_lib.F_V), s.read(_lib.F_EX), s.read(_lib.F_EV)
    o_en = orc.FunnelNeal(scale=w['params'][0])
    cols = np.sort(np.random.RandomState(4).choice(N, size=64, replace=False))
    assert close(EX[cols], o_en.E_val(X[:, cols])[0]) and close(EV, np.sum(V ** 2, axis=0) / 2.)
    E_dev, _ = en.eval(X[:, :4096], want_grad=False)
    assert close(EX[:4096], E_dev)
    o = orc.MarkovJumpHMC(o_en, X0[:, cols], epsilon=eps, beta=beta, num_leapfrog_steps=L, resample=False,
                          rng=orc.PhiloxRNG(11, cols))
    for _ in range(T):
        o.sampling_iteration()
    assert np.array_equal(trans[cols], o.last_transition)
    assert close(X[:, cols], o.state.X) and close(V[:, cols], o.state.V)
    assert close(EX[cols], o.state.EX[0]) and close(s.read(_lib.F_DWELL)[cols], o.dwelling_times)


def test_full_size_c3_product_of_t():
    """configs[2]: ProductOfT, ndims=nbasis=512, nparticles=100000, L=20, float32 on the matrix cores."""
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import ProductOfT
    w = bench.WORKLOADS['c3']
    D, N, L, eps, beta = w['D'], w['N'], w['L'], w['eps'], w['beta']
    W, lognu = bench.pot_model(D)
    X0 = bench.initial_state(w, N, 0)

    class Fixed(ProductOfT):
        def init_X(self):
            self.Xinit = X0
    d = Fixed(ndims=D, nbasis=D, nbatch=N, lognu=lognu, W=W, state_dtype='float32')
    s = MarkovJumpHMC(distribution=d, epsilon=eps, beta=beta, num_leapfrog_steps=L, seed=21, resample=False)
    cols = np.sort(np.random.RandomState(5).choice(N, size=48, replace=False))
    o = orc.MarkovJumpHMC(orc.ProductOfT(W, lognu=lognu, force_dtype=np.float64), X0[:, cols], epsilon=eps, beta=beta,
                          num_leapfrog_steps=L, resample=False, rng=orc.PhiloxRNG(21, cols),
                          state_rounding=lambda a: a.astype(np.float32).astype(np.float64))
    assert np.allclose(s.state.V[:, cols], o.state.V, atol=1e-6)
    resync(s, o, cols)                                      # float32-rounded state on both sides: identical inputs
    d.E_count = d.dEdX_count = 0
    for it in range(3):                                     # the first with every inverse-L cache cold, then warm caches
        n_cold = int(np.sum(~s.state.cache_active))
        e0, g0 = d.E_count, d.dEdX_count
        check_iteration(s, o, delta_rel=2e-5, x_tol=2e-5, e_rtol=2e-5, tag='C3 it %d' % it, cols=cols)
        assert d.E_count - e0 == N + n_cold and d.dEdX_count - g0 == (N + n_cold) * L
        assert s.l_count + s.f_count + s.r_count == (it + 1) * N
        if it < 2:
            resync(s, o, cols)
    # stored energies == energies re-evaluated (device, float32) from the stored state
    Xs = s.state.X[:, :2048]
    assert np.allclose(s.state.EX[0, :2048], d.E(Xs)[0], rtol=2e-5, atol=1e-3)
    cache = s.state.cache_active
    assert np.array_equal(cache, s._dev.read(8) == 0)


def test_full_size_c3_in_the_references_arithmetic():
    """configs[2] as the reference runs it: float64 HMCState arrays around the float32 force (distributions.py:408-415,
    hmc_state.py:29-38) on the tile kernel with the state streamed through its epilogue (csrc/dense_pot64.hip) -- the
    line's top level.  Compared with orc.ProductOfT(force_dtype=float32) WITHOUT any state rounding: what is left is the
    summation order inside the float32 matrix products -- 512-term float32 sums, twice per gradient, 20 gradients per
    trajectory: 4e-6 of the state's scale."""
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import ProductOfT
    w = bench.WORKLOADS['c3f64']
    D, N, L, eps, beta = w['D'], w['N'], w['L'], w['eps'], w['beta']
    W, lognu = bench.pot_model(D)
    X0 = bench.initial_state(w, N, 0)

    class Fixed(ProductOfT):
        def init_X(self):
            self.Xinit = X0
    d = Fixed(ndims=D, nbasis=D, nbatch=N, lognu=lognu, W=W, state_dtype='float64')
    s = MarkovJumpHMC(distribution=d, epsilon=eps, beta=beta, num_leapfrog_steps=L, seed=21, resample=False)
    cols = np.sort(np.random.RandomState(5).choice(N, size=48, replace=False))
    o = orc.MarkovJumpHMC(orc.ProductOfT(W, lognu=lognu, force_dtype=np.float32), X0[:, cols], epsilon=eps, beta=beta,
                          num_leapfrog_steps=L, resample=False, rng=orc.PhiloxRNG(21, cols))
    assert np.allclose(s.state.V[:, cols], o.state.V, rtol=0, atol=1e-13)        # float64 normals, no state rounding
    resync(s, o, cols)
    d.E_count = d.dEdX_count = 0
    for it in range(3):                                     # the first with every inverse-L cache cold, then warm caches
        n_cold = int(np.sum(~s.state.cache_active))
        e0, g0 = d.E_count, d.dEdX_count
        check_iteration(s, o, delta_rel=4e-6, x_tol=4e-6, e_rtol=4e-6, tag='C3 f64 it %d' % it, cols=cols)
        assert d.E_count - e0 == N + n_cold and d.dEdX_count - g0 == (N + n_cold) * L
        assert s.l_count + s.f_count + s.r_count == (it + 1) * N
        if it < 2:
            resync(s, o, cols)
    X = s.state.X
    assert X.dtype == np.float64 and np.abs(X[:, :64] - X[:, :64].astype(np.float32)).max() > 0          # not float32 values
    # stored energies == energies re-evaluated from the stored state (float32 force on the downcast state, float64 sum(V^2))
    assert np.allclose(s.state.EX[0, :2048], d.E(X[:, :2048])[0], rtol=4e-6, atol=1e-4)
    assert np.allclose(s.state.EV[0], np.sum(s.state.V ** 2, axis=0) / 2., rtol=1e-12)
    assert np.array_equal(s.state.cache_active, s._dev.read(8) == 0)


def to_f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


@pytest.mark.parametrize('state', ['float32', 'bfloat16'])
@pytest.mark.parametrize('eps,n_iter', [(0.0625, 1), (0.05, 3)])
def test_full_size_c5_sparse_image_code(eps, n_iter, state):
    """configs[4]: SparseImageCode, 1024 coefficients / 256-pixel patch, nparticles=200000 (the whole batch on one
    GPU), L=25, bf16 matrix-core operands / fp32 accumulate.  `state`: float32 state rows -- the class's default and the
    bench line's `c5`: what the reference's TensorFlow placeholders hold (tf_distributions.py:89), and the form under which
    MarkovJumpHMC keeps its law (tests/test_gpu_stationary.py::test_sic_stationary_law) -- or bfloat16 rows, BASELINE.json's
    wording (`c5bf16` in the line; parity-green and statistically hot: DESIGN.md 3.5).  The oracle rounds its state the same
    way after every commit.  epsilon 2^-4: the mixed-precision restatement of the oracle is exact up to accumulation order
    (see test_sic_iterations_vs_oracle).  epsilon 0.05, the BENCHMARK's own hyper-parameters, three iterations: the kernel
    rounds (step scale x residual) to bf16 where the restatement rounds the residual and scales afterwards -- one more bf16
    rounding per force term, inside the same tolerances."""
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import SparseImageCode
    w = bench.WORKLOADS['c5' if state == 'float32' else 'c5bf16']
    assert w['dtype'] == state
    to_state = to_f32 if state == 'float32' else to_bf16
    N, L, beta = w['N'], w['L'], w['beta']
    assert eps in (0.0625, w['eps'])
    B, y, a0 = bench.sic_model()
    X0 = to_state(bench.initial_state(w, N, 0))
    kw = {} if state == 'float32' else dict(state_dtype='bfloat16')      # float32 rows are the class's default
    d = SparseImageCode(n_patches=1, n_batches=N, cauchy=True, n_basis=1024, basis=B, imgs=y.reshape(256, 1), init=X0, **kw)
    assert d.state_dtype == state
    s = MarkovJumpHMC(distribution=d, epsilon=eps, beta=beta, num_leapfrog_steps=L, seed=31, resample=False)
    cols = np.sort(np.random.RandomState(6).choice(N, size=48, replace=False))
    en = orc.SparseImageCode(B, y.reshape(1, -1), lmbda=0.01, cauchy=True, operand_rounding=to_bf16)
    o = orc.MarkovJumpHMC(en, X0[:, cols], epsilon=eps, beta=beta, num_leapfrog_steps=L, resample=False,
                          rng=orc.PhiloxRNG(31, cols), state_rounding=to_state)
    V0 = s.state.V[:, cols]
    assert np.abs(V0 - o.state.V).max() < (2e-2 if state == 'bfloat16' else 1e-6) and np.array_equal(V0, to_state(V0))
    resync(s, o, cols)
    d.E_count = d.dEdX_count = 0
    for it in range(n_iter):
        n_cold = int(np.sum(~s.state.cache_active))
        e0, g0 = d.E_count, d.dEdX_count
        check_iteration(s, o, delta_rel=5e-4, x_tol=1.0 / 128, e_rtol=5e-4, tag='C5 %s eps %g it %d' % (state, eps, it), cols=cols)
        assert d.E_count - e0 == N + n_cold and d.dEdX_count - g0 == (N + n_cold) * L
        assert s.l_count + s.f_count + s.r_count == (it + 1) * N
        assert np.array_equal(s.state.cache_active, s._dev.read(8) == 0)
        resync(s, o, cols)
