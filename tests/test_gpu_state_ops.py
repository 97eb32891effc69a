"""GPU: the reference's state operators on a host snapshot (HMCState.L / leapfrog / F / FLF / R / update /
cache bookkeeping, hmc_state.py:46-148) and HMCBase.leap_prob (markov_jump_hmc.py:106-114), against the golden
trajectories captured from the reference (G3) and the oracle."""
import numpy as np
import pytest

from oracle import mjhmc_oracle as orc
from tests.helpers import load, bits_equal
from tests.test_gpu_parity import close, product_distribution

pytestmark = pytest.mark.gpu


def _sampler(tag, kind, g):
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    X0, V0 = g[tag + '_X0'], g[tag + '_V0']
    eps, L = float(g[tag + '_hp'][0]), int(g[tag + '_hp'][1])
    fake = dict(kind=kind, par_sigma=1.3, par_scale1=100, par_scale2=4,
                par_conditioning=10 ** np.linspace(-2, 0, X0.shape[0]))
    return MarkovJumpHMC(distribution=product_distribution(fake, X0), epsilon=eps, beta=0.3, num_leapfrog_steps=L,
                         Vinit=V0, seed=3)


@pytest.mark.parametrize('tag,kind', [('iso_2x100', 'iso'), ('iso_512x32', 'iso'), ('diag_16x24', 'diag'),
                                      ('rough_4x16', 'rough')])
def test_L_and_FLF_against_the_reference_trajectories(tag, kind):
    g = load('g3_trajectories')
    s = _sampler(tag, kind, g)
    d = s.distribution
    n = g[tag + '_X0'].shape[1]
    e0, g0 = d.E_count, d.dEdX_count
    Z = s.state.copy().L()
    assert (d.E_count - e0, d.dEdX_count - g0) == (n, n * s.num_leapfrog_steps)
    assert close(Z.X, g[tag + '_L_X']) and close(Z.V, g[tag + '_L_V']) and close(Z.dEdX, g[tag + '_L_g'])
    assert close(Z.EX[0], g[tag + '_L_EX']) and close(Z.EV[0], g[tag + '_L_EV'])
    if kind == 'diag':                       # exact products: the literal operation order reproduces NumPy's bits
        assert bits_equal(Z.X, g[tag + '_L_X']) and bits_equal(Z.V, g[tag + '_L_V'])
    W = s.state.copy().FLF()
    assert close(W.X, g[tag + '_FLF_X']) and close(W.V, g[tag + '_FLF_V']) and close(W.dEdX, g[tag + '_FLF_g'])
    assert close(W.EX[0], g[tag + '_FLF_EX']) and close(W.EV[0], g[tag + '_FLF_EV'])
    # L then F then L then F is the identity up to rounding (time reversibility of the integrator)
    back = s.state.copy().L().F().L().F()
    assert np.allclose(back.X, g[tag + '_X0'], rtol=1e-9, atol=1e-9 * np.abs(g[tag + '_X0']).max()) or kind == 'rough'


def test_single_steps_compose_to_L_and_bookkeeping_operators():
    g = load('g3_trajectories')
    s = _sampler('diag_16x24', 'diag', g)
    Z = s.state.copy()
    for _ in range(s.num_leapfrog_steps):
        Z.leapfrog()
    assert bits_equal(Z.X, g['diag_16x24_L_X']) and bits_equal(Z.V, g['diag_16x24_L_V'])
    assert bits_equal(Z.EX, s.state.EX)                       # leapfrog() leaves the energies alone, like the reference
    Z.update_EV()
    Z.update_EX()
    assert close(Z.EX[0], g['diag_16x24_L_EX']) and close(Z.EV[0], g['diag_16x24_L_EV'])
    # operators act on active_idx only
    A = s.state.copy()
    A.active_idx = np.array([1, 5, 7])
    A.L()
    rest = np.setdiff1d(np.arange(A.nbatch), [1, 5, 7])
    assert bits_equal(A.X[:, rest], g['diag_16x24_X0'][:, rest])
    assert bits_equal(A.X[:, [1, 5, 7]], g['diag_16x24_L_X'][:, [1, 5, 7]])
    # F, update, cache bookkeeping
    B = s.state.copy()
    assert bits_equal(B.copy().F().V, -B.V)
    B.update(np.array([0, 2]), Z)
    assert bits_equal(B.X[:, [0, 2]], Z.X[:, [0, 2]]) and bits_equal(B.X[:, 1], g['diag_16x24_X0'][:, 1])
    B.cache_flf_state(np.array([3]), Z)
    assert B.cache_active[3] and not B.cache_active[4]
    B.clear_flf_cache(np.array([3]))
    assert not B.cache_active.any()
    # R: the reference's expression on the process-global NumPy stream
    C = s.state.copy()
    np.random.seed(12)
    C.R()
    np.random.seed(12)
    want = s.state.V * np.sqrt(1. - s.beta) + np.random.randn(*C.V.shape) * np.sqrt(s.beta)
    assert bits_equal(C.V, want) and close(C.EV[0], np.sum(want ** 2, axis=0) / 2.)


def test_operators_on_the_live_state_move_the_sampler():
    """In the reference `sampler.state` IS the HMCState (hmc_state.py:46-148): `sampler.state.L()` moves the sampler.  The
    device view offers the same operators: snapshot, operate, write back."""
    g = load('g3_trajectories')
    s = _sampler('diag_16x24', 'diag', g)
    s.state.L()
    assert bits_equal(s.state.X, g['diag_16x24_L_X']) and bits_equal(s.state.V, g['diag_16x24_L_V'])
    assert close(s.state.EX[0], g['diag_16x24_L_EX']) and close(s.state.EV[0], g['diag_16x24_L_EV'])
    assert not s.state.cache_active.any()
    s.state.F()
    assert bits_equal(s.state.V, -g['diag_16x24_L_V'])
    s.state.L().F()                                             # F L F L = identity up to rounding
    assert np.allclose(s.state.X, g['diag_16x24_X0'], rtol=1e-9, atol=1e-9)
    # cache bookkeeping on the device: H of the cached inverse-L state, NaN = cold
    Z = s.state.copy().L()
    s.state.cache_flf_state(np.array([2, 5]), Z)
    ca = s.state.cache_active
    assert ca[2] and ca[5] and ca.sum() == 2 and close(s.state.H_flf[0, [2, 5]], Z.H()[0, [2, 5]])
    s.state.clear_flf_cache(np.array([5]))
    assert s.state.cache_active.sum() == 1
    # update: columns of another state; R: the reference's expression on the process-global NumPy stream
    X_before = s.state.X
    s.state.update(np.array([0, 3]), Z)
    assert bits_equal(s.state.X[:, [0, 3]], Z.X[:, [0, 3]]) and bits_equal(s.state.X[:, 1], X_before[:, 1])
    V_before = s.state.V
    np.random.seed(3)
    s.state.R()
    np.random.seed(3)
    want = V_before * np.sqrt(1. - s.beta) + np.random.randn(*V_before.shape) * np.sqrt(s.beta)
    assert bits_equal(s.state.V, want)
    s.sampling_iteration()                                      # and the sampler carries on from there
    assert s.l_count + s.f_count + s.r_count == s.nbatch


def test_leap_prob_and_transition_rates():
    g = load('g3_trajectories')
    s = _sampler('iso_2x100', 'iso', g)
    Z1 = s.state.copy()
    Z2 = s.state.copy().L().F()
    Ediff = Z1.H() - Z2.H()
    p = s.leap_prob(Z1, Z2)
    assert p.shape == (1, 100) and np.all(p[Ediff >= 0] == 1) and np.allclose(p[Ediff < 0], np.exp(Ediff[Ediff < 0]))
    assert np.allclose(s.transition_rates(Z1, Z2), np.exp(Ediff) ** .5)


def test_dense_energies_integrate_snapshots_too():
    """figures/poe_fig.py:58-76 integrates snapshots of a ProductOfT sampler (details against the oracle:
    tests/test_gpu_dense_parity.py::test_pot_state_assignment_and_leapfrog_operator)."""
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import ProductOfT
    X0 = np.random.RandomState(0).randn(36, 40)

    class Fixed(ProductOfT):
        def init_X(self):
            self.Xinit = X0
    d = Fixed(ndims=36, nbasis=36, nbatch=40, state_dtype='float32')
    s = MarkovJumpHMC(distribution=d, epsilon=0.1, seed=1, resample=False, num_leapfrog_steps=4)
    before = (d.E_count, d.dEdX_count)
    Z = s.state.copy().L()
    assert Z.X.shape == (36, 40) and np.isfinite(Z.X).all() and not np.array_equal(Z.X, s.state.X)
    assert (d.E_count - before[0], d.dEdX_count - before[1]) == (40, 4 * 40)
    assert np.allclose(Z.EX[0], d.E_val(Z.X)[0], rtol=2e-5, atol=1e-4)


def test_cached_init_X_and_load_cache(tmp_path):
    """Distribution.cached_init_X / load_cache (distributions.py:104-149, 182-195): generate once (toy step counts),
    then serve the MJHMC end points to continuous-time samplers and ControlHMC's to the others."""
    from mjhmc_amd.misc.distributions import Gaussian
    d = Gaussian(ndims=4, nbatch=30, log_conditioning=1)
    d.max_n_particles = 64
    with pytest.raises(IOError):
        d.load_cache(str(tmp_path))
    d.mjhmc = True
    d.cached_init_X(str(tmp_path), burn_in_steps=300, var_steps=100, seed=5)
    mj, emc_var, true_var, ctl = d.load_cache(str(tmp_path))
    assert mj.shape == (4, 64) and ctl.shape == (4, 64) and emc_var > 0 and true_var > 0
    assert d.nbatch == 30 and not d.generation_instance and np.array_equal(d.Xinit, mj[:, :30])
    d.mjhmc = False
    d.cached_init_X(str(tmp_path))                      # second call only reads the file
    assert np.array_equal(d.Xinit, ctl[:, :30])


@pytest.mark.parametrize('kind', ['diag', 'pot', 'control'])
def test_save_and_load_state_continue_bit_for_bit(kind, tmp_path):
    """.npz checkpoint of a sampler (state, inverse-L cache, RNG tick, counters): a fresh sampler that loads it continues
    exactly like the original."""
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC, ControlHMC
    from mjhmc_amd.misc.distributions import Gaussian, ProductOfT
    X0 = np.random.RandomState(3).randn(36, 70)

    def make():
        if kind == 'pot':
            class FixedT(ProductOfT):
                def init_X(self):
                    self.Xinit = X0
            rs = np.random.RandomState(8)
            W = rs.randn(36, 36) * (rs.rand(36, 36) < 0.05) + np.eye(36)
            d = FixedT(ndims=36, nbasis=36, nbatch=70, W=W, lognu=np.log(rs.rand(36) * 2 + 2.1))
        else:
            class FixedG(Gaussian):
                def init_X(self):
                    self.Xinit = X0
            d = FixedG(ndims=36, nbatch=70, log_conditioning=2)
        cls, kw = (ControlHMC, {}) if kind == 'control' else (MarkovJumpHMC, dict(resample=False))
        return cls(distribution=d, epsilon=0.3, beta=0.4, num_leapfrog_steps=5, seed=77, **kw), d

    a, da = make()
    a.sample(9)                                             # fused / batched launches
    a.sampling_iteration()
    path = str(tmp_path / 'ckpt.npz')
    a.save_state(path)
    want = a.sample(6)
    b, db = make()
    b.load_state(path)
    assert np.array_equal(b.state.X, np.load(path)['X']) and np.array_equal(b.state.cache_active, ~np.isnan(np.load(path)['H_flf']))
    got = b.sample(6)
    assert np.array_equal(got, want)
    assert np.array_equal(a.state.X, b.state.X) and np.array_equal(a.state.V, b.state.V)
    assert np.array_equal(a.state.EX, b.state.EX) and np.array_equal(a.state.EV, b.state.EV)
    assert (a.l_count, a.f_count, a.r_count, a.fl_count) == (b.l_count, b.f_count, b.r_count, b.fl_count)
    assert (da.E_count, da.dEdX_count) == (db.E_count, db.dEdX_count)


def test_tf_distributions_import_path():
    """`from mjhmc.misc.tf_distributions import TFGaussian, Funnel, SparseImageCode` keeps working with the package name
    swapped; TFGaussian samples like TestGaussian (same energy)."""
    from mjhmc_amd.misc.tf_distributions import TFGaussian, Funnel, SparseImageCode   # noqa: F401
    from mjhmc_amd.misc.distributions import TestGaussian
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    outs = []
    for cls in (TFGaussian, TestGaussian):
        np.random.seed(4)
        s = MarkovJumpHMC(distribution=cls(ndims=5, nbatch=60, sigma=2.0), epsilon=0.3, beta=0.2, num_leapfrog_steps=4, seed=6)
        np.random.seed(5)
        outs.append(s.sample(6))
    assert bits_equal(outs[0], outs[1])


@pytest.mark.parametrize('what', ['iso', 'iso_wide', 'iso_wide32', 'pot32', 'pot64', 'pot64_L0', 'host'])
def test_rollback_undoes_a_committed_single_iteration(what):
    """mjhmc_rollback (include/mjhmc_hip.h): after mjhmc_iterate(1) committed, the pre-move state comes back bit for bit
    -- by flipping the ping-pong parities (register / tile kernels) or by copying back the rows the multi-pass commit
    left in its proposal workspace (rows wider than the register kernels, ProductOfT float64 with L = 0, opaque
    callables).  A second rollback, and one after a multi-iteration call, are refused."""
    from mjhmc_amd import engine, _lib
    from tests.helpers import ref_init_weights
    ctx = engine.context(0)
    rs = np.random.RandomState(3)
    L = 4
    if what.startswith('iso'):
        D, N = (1030, 50) if what.startswith('iso_wide') else (24, 200)
        en = engine.DeviceEnergy(ctx, _lib.E_ISO_GAUSS, D, [1.3])
        # iso_wide32: float32 state on rows only the multi-pass path holds (a float64 sampler rounding its state, Shape::round32)
        s = engine.DeviceSampler(en, rs.randn(D, N), seed=5, dtype='float32' if what == 'iso_wide32' else 'float64')
    elif what.startswith('pot'):
        D, N = 36, 100
        W, lognu = ref_init_weights(D, D)
        en = engine.DeviceEnergy(ctx, _lib.E_PRODUCT_OF_T, D, np.concatenate([[float(D)], (W + np.eye(D)).ravel(), np.exp(lognu), np.zeros(D)]))
        s = engine.DeviceSampler(en, rs.randn(D, N), seed=5, dtype='float32' if what == 'pot32' else 'float64')
        if what == 'pot64_L0':
            L = 0
    else:
        D, N = 12, 40
        A = rs.randn(D, D)
        A = A.dot(A.T) / D + np.eye(D)
        en = engine.DeviceEnergy.host(ctx, D, lambda X: 0.5 * np.sum(X * A.dot(X), axis=0), lambda X: A.dot(X))
        s = engine.HostEnergySampler(en, rs.randn(D, N), seed=5)
    s.set_hparams(0.1, L, 0.2, 1.0)
    s.iterate(2)
    fields = [_lib.F_X, _lib.F_V, _lib.F_EX, _lib.F_EV, _lib.F_HFLF] + ([_lib.F_DEDX] if what != 'iso' else [])
    before = [s.read(f) for f in fields]
    tick = s.get_tick()
    st, done = s.iterate(1)
    assert done == 1
    assert any(not np.array_equal(a, s.read(f), equal_nan=True) for a, f in zip(before, fields))
    s.rollback()
    for a, f in zip(before, fields):
        assert np.array_equal(a, s.read(f), equal_nan=True), (what, f)
    assert s.get_tick() == tick + 1                         # the tick stays consumed
    with pytest.raises(_lib.EngineError):
        s.rollback()
    # the sampler goes on from the restored state exactly like one that never made the rolled-back attempt
    s.iterate(2)
    if what != 'host':                                      # (a host-evaluated energy is driven one attempt at a time)
        with pytest.raises(_lib.EngineError):
            s.rollback()                                    # a multi-iteration call cannot be undone this way
    s.close()


# ---------------------------------------------------------------------------------------------
# sample(): every state goes to the host while the next iterations run (mjhmc_iterate_download), through a ring that
# may be smaller than the run
# ---------------------------------------------------------------------------------------------
def _sampler_for(what, seed=11):
    from mjhmc_amd.samplers import markov_jump_hmc as M
    from mjhmc_amd.misc import distributions as Dm
    from tests.helpers import ref_init_weights
    rs = np.random.RandomState(4)
    if what == 'iso_fused':                 # Gaussian force: fused launches write the ring from inside the launch
        D, N = 24, 5000
        X0 = rs.randn(D, N)

        class Fixed(Dm.TestGaussian):
            def gen_init_X(self):
                self.Xinit = X0
        d = Fixed(ndims=D, nbatch=N, sigma=1.2)
        kw = dict(epsilon=0.2, num_leapfrog_steps=5)
    elif what == 'funnel_compacted':        # a funnel, big batch: the fused row kernel (two-lane groups, short rows) writes the ring
                                            # (until the row form: a trajectory + a jump-process launch per iteration)
        D, N = 10, 170000
        X0 = rs.randn(D, N)

        class Fixed(Dm.Funnel):
            def gen_init_X(self):
                self.Xinit = X0
        d = Fixed(ndims=D, nbatch=N, scale=1.5)
        kw = dict(epsilon=0.1, num_leapfrog_steps=5)
    else:                                   # ProductOfT on the tile kernels, two free-running parts
        D, N = 36, 8000
        W, lognu = ref_init_weights(D, D)
        X0 = rs.randn(D, N)

        class Fixed(Dm.ProductOfT):
            def init_X(self):
                self.Xinit = X0
        d = Fixed(ndims=D, nbatch=N, lognu=lognu, W=W + np.eye(D), state_dtype='float64' if what == 'pot64' else 'float32')
        kw = dict(epsilon=0.1, num_leapfrog_steps=6)
    return d, kw


@pytest.mark.parametrize('what', ['iso_fused', 'funnel_compacted', 'pot32', 'pot64'])
@pytest.mark.parametrize('cls_name', ['MarkovJumpHMC', 'ControlHMC'])
def test_streamed_sample_equals_the_recorded_ring(what, cls_name, monkeypatch):
    from mjhmc_amd.samplers import markov_jump_hmc as M
    n = 7
    outs = []
    for variant in ('streamed', 'ring of 3 slots', 'recorded, then read'):
        d, kw = _sampler_for(what)
        extra = dict(resample=False) if cls_name == 'MarkovJumpHMC' else {}
        s = getattr(M, cls_name)(distribution=d, seed=21, beta=0.2, **extra, **kw)
        if variant == 'recorded, then read':
            monkeypatch.setattr(type(s), '_streams', lambda self, po, rp: False)
        elif variant == 'ring of 3 slots':
            monkeypatch.setattr(type(s._dev), 'ring_budget_slots', lambda self, n_wanted, share=0.6, staging=True: min(n_wanted, 3))
        outs.append((s.sample(n), s.state.X, s.l_count, s.f_count, s.r_count, d.E_count, d.dEdX_count, s.eval_trace(n).tolist()))
        assert len(outs[-1][-1]) == n, 'eval_trace() describes the whole batch call, chunked or not'
        monkeypatch.undo()
    for o in outs[1:]:
        if not bits_equal(outs[0][0], o[0]):
            bad = np.argwhere(outs[0][0] != o[0])
            Nn = outs[0][1].shape[1]
            raise AssertionError(('samples differ', len(bad), 'first', bad[:6].tolist(), 'time slots', sorted(set((bad[:, 1] // Nn).tolist())),
                                  'particles', sorted(set((bad[:, 1] % Nn).tolist()))[:12], 'a', outs[0][0][tuple(bad[0])], 'b', o[0][tuple(bad[0])]))
        assert bits_equal(outs[0][0], o[0]) and bits_equal(outs[0][1], o[1]) and outs[0][2:] == o[2:]
    assert outs[0][0].shape[1] == n * outs[0][1].shape[1]
    assert bits_equal(outs[0][0][:, -outs[0][1].shape[1]:], outs[0][1])        # the last sample is the live state


def test_resampling_on_the_host_when_the_ring_does_not_fit(monkeypatch):
    """sample() with dwell-time resampling needs all n + 1 states at once; when they exceed the device they are streamed
    through a smaller ring and the columns are picked on the host: same uniforms, same indices, same columns."""
    from mjhmc_amd.samplers import markov_jump_hmc as M
    outs = []
    for small in (False, True):
        d, kw = _sampler_for('pot32')
        s = M.MarkovJumpHMC(distribution=d, seed=21, beta=0.2, **kw)
        if small:
            monkeypatch.setattr(type(s._dev), 'ring_budget_slots', lambda self, n_wanted, share=0.6, staging=True: min(n_wanted, 4))
        np.random.seed(3)
        outs.append((s.sample(9), s._last_resample_idx.copy(), s.dwelling_times.copy()))
        monkeypatch.undo()
    assert np.array_equal(outs[0][1], outs[1][1]) and bits_equal(outs[0][0], outs[1][0]) and bits_equal(outs[0][2], outs[1][2])


def test_ring_request_beyond_the_device_keeps_the_ring_and_says_sizes():
    from mjhmc_amd import engine, _lib
    ctx = engine.context(0)
    en = engine.DeviceEnergy(ctx, _lib.E_ISO_GAUSS, 512, [1.0])
    s = engine.DeviceSampler(en, np.random.RandomState(0).randn(512, 4096), seed=1)
    s.set_hparams(0.1, 3, 0.05, 1.0)
    s.ring_alloc(4)
    s.iterate(4, ring_slot0=0)
    before = s.ring_read(0, 4)
    free, total = ctx.mem_info()
    assert 0 < free <= total
    too_many = int(2 * total // (512 * 4096 * 8)) + 8
    with pytest.raises(_lib.EngineError, match=r'does not fit the device .* the ring the sampler had is kept'):
        s.ring_alloc(too_many)
    assert bits_equal(before, s.ring_read(0, 4))
    st, done = s.iterate(2, ring_slot0=2)                           # and is still usable
    assert done == 2
    s.close()
