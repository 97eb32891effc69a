"""GPU: the device autocorrelation (mjhmc_ring_autocor / mjhmc_autocor through the C ABI) against the
NumPy restatement of mjhmc/misc/autocor.py:37-117,177-211.

Floating-point bar: the reference itself compares its two estimators at 1e-7 (tests/test_fast_ac.py:14);
both sides here are float64 transforms of the same data, so the test asks for 1e-10 absolute on the
normalised autocorrelation (values in [-1, 1])."""
import numpy as np
import pytest

from oracle import autocor_oracle as aco

pytestmark = pytest.mark.gpu
ATOL = 1e-10


@pytest.mark.parametrize('D,N,T', [(3, 50, 37), (2, 100, 64), (1, 1, 5), (5, 33, 1000), (4, 9, 1)])
def test_host_array_estimators_match_oracle(D, N, T):
    from mjhmc_amd.misc import autocor as ac
    rs = np.random.RandomState(D * 1000 + T)
    # AR(1)-like series so that the autocorrelation is not just noise
    x = np.zeros((D, N, T))
    e = rs.randn(D, N, T)
    x[:, :, 0] = e[:, :, 0]
    for t in range(1, T):
        x[:, :, t] = 0.9 * x[:, :, t - 1] + e[:, :, t]
    np.testing.assert_allclose(ac.fft_autocor(x), aco.fft_autocor(x), rtol=0, atol=ATOL)
    tr = np.arange(T, dtype=float)
    if T >= 3:
        got, _, _ = ac.slow_autocorrelation(x, tr, tr, half_window=False)
        want, _, _ = aco.slow_autocorrelation(x, tr, tr, half_window=False)
        np.testing.assert_allclose(got, want, rtol=0, atol=ATOL)
    if T >= 6:
        for half in (True, False):
            got, ge, gg = ac.autocorrelation(x, tr, tr, half_window=half, brute_force=True)
            want, we, wg = aco.autocorrelation(x, tr, tr, half_window=half, brute_force=True)
            assert got.shape == want.shape and np.array_equal(ge, we) and np.array_equal(gg, wg)
            np.testing.assert_allclose(got, want, rtol=0, atol=ATOL)
        got, _, _ = ac.slow_autocorrelation(x, tr, tr, half_window=True)
        want, _, _ = aco.slow_autocorrelation(x, tr, tr, half_window=True)
        np.testing.assert_allclose(got, want, rtol=0, atol=ATOL)
    got, ge, gg = ac.autocorrelation(x, tr, tr, half_window=False)             # the default (fft) branch
    np.testing.assert_allclose(got, aco.fft_autocor(x), rtol=0, atol=ATOL)
    cached = ac.autocorrelation(x, tr, tr, half_window=False, brute_force=True, normalize=False, cached_var=2.5)[0] \
        if T >= 6 else None
    if cached is not None:
        assert cached[0, 0] == 2.5


def _gaussian(D, N, seed):
    from mjhmc_amd.misc.distributions import Gaussian
    X0 = np.random.RandomState(seed).randn(D, N)

    class Fixed(Gaussian):
        def init_X(self):
            self.Xinit = X0
    return Fixed(ndims=D, nbatch=N, log_conditioning=2)


@pytest.mark.parametrize('cls_name,D,N,T', [('MarkovJumpHMC', 10, 60, 32), ('ControlHMC', 7, 130, 32),
                                            ('MarkovJumpHMC', 512, 70, 32), ('MarkovJumpHMC', 10, 60, 48)])   # T = 48: the transform pipeline from the ring
def test_ring_resident_autocorrelation(cls_name, D, N, T, monkeypatch):
    """calculate_autocorrelation never downloads the samples; a twin sampler's downloaded samples through
    the oracle give the same curve, and the counter traces are those of generate_samples."""
    from mjhmc_amd.samplers import markov_jump_hmc as M
    from mjhmc_amd.misc.autocor import calculate_autocorrelation, generate_samples
    kw = dict(epsilon=0.4, beta=0.3, num_leapfrog_steps=5, seed=31)
    if cls_name == 'MarkovJumpHMC':
        kw['resample'] = False
    cls = getattr(M, cls_name)
    ac, e, g = calculate_autocorrelation(cls, _gaussian(D, N, 3), num_steps=T, **kw)
    samples, e2, g2 = generate_samples(cls, _gaussian(D, N, 3), num_steps=T, **kw)
    assert ac.shape == (T,) and np.array_equal(e, e2) and np.array_equal(g, g2)
    np.testing.assert_allclose(ac, aco.fft_autocor(samples), rtol=0, atol=ATOL)
    # a staging budget of 1 MB forces several chunks of series (and a ragged last one): the test build of the library
    # reads MJHMC_AUTOCOR_STAGING_MB; host-array source and ring source against the product library's one-chunk result
    from mjhmc_amd import engine, _lib
    from tests.helpers import hooks_context
    monkeypatch.setenv('MJHMC_AUTOCOR_STAGING_MB', '1')
    monkeypatch.setenv('MJHMC_AUTOCOR_TRANSFORM', '1')      # (rings of <= 32 samples: the product sums the lag products directly)
    hctx = hooks_context(0)
    for linear in (False, True):
        want = engine.context(0).autocor(samples, linear=linear)
        got = hctx.autocor(samples, linear=linear)
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-13 * abs(want[0]))
    rs = np.random.RandomState(9)
    Xr = rs.randn(D, N)
    rings = []
    for ctx in (engine.context(0), hctx):
        en = engine.DeviceEnergy(ctx, _lib.E_ISO_GAUSS, D, [1.0])
        smp = engine.DeviceSampler(en, Xr, seed=3)
        smp.set_hparams(0.3, 4, 0.1, 1.0)
        smp.ring_alloc(T)
        smp.iterate(T, ring_slot0=0)
        rings.append(smp.ring_autocor(0, T))
        smp.close()
    np.testing.assert_allclose(rings[1], rings[0], rtol=1e-12, atol=1e-12 * abs(rings[0][0]))
    monkeypatch.delenv('MJHMC_AUTOCOR_STAGING_MB', raising=False)
    monkeypatch.delenv('MJHMC_AUTOCOR_TRANSFORM', raising=False)
    # gradient-budget form: the curve of the truncated run
    ac3, e3, g3 = calculate_autocorrelation(cls, _gaussian(D, N, 3), num_grad_steps=60, **kw)
    k = int(np.nonzero(g2 >= 60)[0][0]) + 1 if np.any(g2 >= 60) else None
    if k is not None:
        assert ac3.shape == (k,)
        np.testing.assert_allclose(ac3, aco.fft_autocor(samples[:, :, :k]), rtol=0, atol=ATOL)


def test_resampling_sampler_goes_through_the_host_array_path():
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.autocor import calculate_autocorrelation, generate_samples
    kw = dict(epsilon=0.4, beta=0.3, num_leapfrog_steps=5, seed=31)
    np.random.seed(11)
    ac, e, g = calculate_autocorrelation(MarkovJumpHMC, _gaussian(4, 40, 8), num_steps=20, **kw)
    np.random.seed(11)
    samples, e2, g2 = generate_samples(MarkovJumpHMC, _gaussian(4, 40, 8), num_steps=20, **kw)
    assert np.array_equal(e, e2) and np.array_equal(g, g2)
    np.testing.assert_allclose(ac, aco.fft_autocor(samples), rtol=0, atol=ATOL)


def test_float32_state_ring():
    """ProductOfT keeps float32 state: the ring is float32, the transform is float64 of those values."""
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import ProductOfT
    rs = np.random.RandomState(2)
    W = (rs.randn(36, 36) * (rs.rand(36, 36) <= 0.2) + np.eye(36)).astype(np.float32)
    X0 = rs.randn(36, 45)

    class Fixed(ProductOfT):
        def init_X(self):
            self.Xinit = X0
    d = Fixed(ndims=36, nbasis=36, nbatch=45, W=W, lognu=np.log(rs.rand(36) * 2 + 2.1), b=np.zeros(36), state_dtype='float32')
    smp = MarkovJumpHMC(distribution=d, epsilon=0.1, beta=0.2, num_leapfrog_steps=4, seed=5, resample=False)
    T = 32
    smp._record(T)
    samples = smp._stack(T, True)
    sums = smp._dev.ring_autocor(0, T)
    np.testing.assert_allclose(sums / sums[0], aco.fft_autocor(samples), rtol=0, atol=ATOL)
    lin = smp._dev.ring_autocor(0, T, linear=True)
    np.testing.assert_allclose(lin, aco.linear_lag_sums(samples), rtol=1e-11)
    sub = smp._dev.ring_autocor(8, 16)                                       # a window of the ring
    np.testing.assert_allclose(sub, aco.circular_lag_sums(samples[:, :, 8:24]), rtol=1e-11)


def test_bad_arguments_fail_loudly():
    from mjhmc_amd._lib import EngineError
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    smp = MarkovJumpHMC(distribution=_gaussian(3, 10, 1), epsilon=0.3, seed=1, resample=False)
    smp._record(4)
    with pytest.raises(EngineError):
        smp._dev.ring_autocor(2, 3)                                          # runs past the ring
    with pytest.raises(EngineError):
        smp._dev.ring_autocor(0, 0)


# ---------------------------------------------------------------------------------------------
# G9: the reference's own generate_samples / fft_autocor / slow_autocorrelation outputs (captured by executing those
# functions from the reference's file, oracle/capture_golden.py: capture_autocor) on a recorded run
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['g9_generate_mjhmc_diag_6x40', 'g9_generate_control_iso_3x50'])
def test_g9_generate_samples_and_autocorrelation_match_the_reference(name):
    from mjhmc_amd.samplers import markov_jump_hmc as M
    from mjhmc_amd.misc import autocor as ac
    from tests.helpers import load, bits_equal
    from tests.test_gpu_parity import product_distribution
    from tests.test_oracle_golden import g9_replay_feed
    g = load(name)
    T = int(g['T'])
    cls = getattr(M, str(g['cls']))
    kw = dict(epsilon=float(g['eps']), beta=float(g['beta_in']), num_leapfrog_steps=int(g['L']), Vinit=g['normals'][0], seed=5)
    if str(g['cls']) == 'MarkovJumpHMC':
        kw['resample'] = False
    samples, e_evals, grad_evals = ac.generate_samples(cls, product_distribution(g, g['Xinit']), num_steps=T,
                                                       replay=g9_replay_feed(g), **kw)
    assert samples.shape == g['samples'].shape
    assert np.array_equal(e_evals, g['e_evals']) and np.array_equal(grad_evals, g['grad_evals'])   # counter traces: exact
    np.testing.assert_allclose(samples, g['samples'], rtol=1e-10, atol=1e-12)
    if str(g['kind']) == 'diag':
        assert bits_equal(samples, g['samples'])            # exact-product force: the chain is bit-identical
    # the autocorrelation estimators on the REFERENCE's samples, and end to end on the device ring
    np.testing.assert_allclose(ac.fft_autocor(g['samples']), g['fft_autocor'], rtol=0, atol=ATOL)
    got, _, _ = ac.slow_autocorrelation(g['samples'], g['e_evals'], g['grad_evals'], half_window=False)
    np.testing.assert_allclose(got, g['slow_autocor'], rtol=0, atol=ATOL)
    auto, e2, g2 = ac.calculate_autocorrelation(cls, product_distribution(g, g['Xinit']), num_steps=T,
                                                replay=g9_replay_feed(g), **kw)
    np.testing.assert_allclose(auto, g['fft_autocor'], rtol=0, atol=ATOL)
    assert np.array_equal(e2, g['e_evals']) and np.array_equal(g2, g['grad_evals'])


@pytest.mark.parametrize('T', [1, 2, 5, 8, 9, 16, 17, 31, 32])
@pytest.mark.parametrize('dtype,D', [('float64', 24), ('float32', 36)])
def test_short_rings_sum_the_lag_products_directly(T, dtype, D):
    """Rings of up to 32 samples (what a BASELINE-size batch holds) are not transformed: lag_sums_direct reads the ring once
    and adds the linear lag products per series (csrc/autocor.hip); circular sums = lin[k] + lin[T - k].  Against the
    oracle's lag sums of the downloaded ring, linear and circular, ragged N and every T that meets a template boundary."""
    from mjhmc_amd import engine, _lib
    from tests.helpers import ref_init_weights
    ctx = engine.context(0)
    N = 333
    rs = np.random.RandomState(T + D)
    if dtype == 'float64':
        en = engine.DeviceEnergy(ctx, _lib.E_ISO_GAUSS, D, [1.0])
    else:
        W, lognu = ref_init_weights(D, D)
        en = engine.DeviceEnergy(ctx, _lib.E_PRODUCT_OF_T, D, np.concatenate([[float(D)], W.ravel(), np.exp(lognu), np.zeros(D)]))
    smp = engine.DeviceSampler(en, rs.randn(D, N), seed=3, dtype=dtype)
    smp.set_hparams(0.2, 4, 0.1, 1.0)
    smp.ring_alloc(T + 2)
    smp.iterate(T + 2, ring_slot0=0)
    samples = smp.ring_read(1, T, stacked=True)                       # a sub-range of the ring's slots
    lin = smp.ring_autocor(1, T, linear=True)
    circ = smp.ring_autocor(1, T)
    np.testing.assert_allclose(lin, aco.linear_lag_sums(samples), rtol=1e-12)
    np.testing.assert_allclose(circ, aco.circular_lag_sums(samples), rtol=1e-12, atol=1e-12 * abs(circ[0]))
    smp.close()
