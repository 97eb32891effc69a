"""GPU: the HIP engine (through the C ABI) against the golden vectors and the NumPy oracle.

Bars: integer/index bookkeeping (transitions, cache flags, counters, evaluation counts, resample
indices) bit-exact; float64 state and energies within 1e-10 relative (RTOL below) -- and X, V
bit-identical wherever the force is an exact product (sigma = 1 isotropic / diagonal Gaussian),
because the kernels are built with -ffp-contract=off and follow the reference's operation order.
"""
import os

import numpy as np
import pytest

from oracle import mjhmc_oracle as orc
from tests.helpers import load, oracle_energy, bits_equal

pytestmark = pytest.mark.gpu
RTOL = 1e-10
np.seterr(all='ignore')


def close(a, b, scale=None):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if scale is None:
        fin = np.abs(b[np.isfinite(b)])
        scale = float(fin.max()) if fin.size else 1.0
    return a.shape == b.shape and np.allclose(a, b, rtol=RTOL, atol=RTOL * 1e-2 * max(scale, 1e-300), equal_nan=True)


def product_distribution(g, Xinit):
    from mjhmc_amd.misc import distributions as D
    kind = str(g['kind'])
    ndims, nbatch = Xinit.shape
    if kind == 'iso':
        base, kw = D.TestGaussian, dict(sigma=float(g['par_sigma']))
    elif kind == 'diag':
        base, kw = D.Gaussian, dict(log_conditioning=2)
    elif kind == 'rough':
        base, kw = D.RoughWell, dict(scale1=int(g['par_scale1']), scale2=int(g['par_scale2']))
    elif kind == 'mm':
        base, kw = D.MultimodalGaussian, dict(separation=int(g['par_separation']))
    else:
        raise KeyError(kind)

    class Fixed(base):
        def init_X(self):
            self.Xinit = Xinit

    d = Fixed(ndims=ndims, nbatch=nbatch, **kw)
    if kind == 'diag':
        d.conditioning = g['par_conditioning']
        d.J = np.diag(d.conditioning)
        d._dev = None
    return d


# ---------------------------------------------------------------------------------------------
# G2: single energy / gradient evaluations
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag,kind', [('iso_2x100', 'iso'), ('iso_512x64', 'iso'), ('diag_10x33', 'diag'),
                                      ('rough_5x40', 'rough'), ('mm_3x20', 'mm')])
def test_g2_energy_evaluations(tag, kind):
    from mjhmc_amd.misc import distributions as D
    g = load('g2_energies')
    X = g[tag + '_X']
    nd, n = X.shape
    d = {'iso': lambda: D.TestGaussian(nd, n, sigma=1.3), 'diag': lambda: D.Gaussian(nd, n, log_conditioning=6),
         'rough': lambda: D.RoughWell(nd, n), 'mm': lambda: D.MultimodalGaussian(nd, n)}[kind]()
    E = d.E(X)
    G = d.dEdX(X)
    assert E.shape == (1, n) and G.shape == (nd, n)
    assert (d.E_count, d.dEdX_count) == (n, n)
    assert close(E[0], g[tag + '_E'])
    assert close(G, g[tag + '_g'])
    if kind == 'diag':
        assert bits_equal(G, g[tag + '_g'])          # j*x is the exact product the reference forms


# ---------------------------------------------------------------------------------------------
# G3: leapfrog trajectories, observed through an iteration whose L clock is forced to win
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('tag,kind', [('iso_2x100', 'iso'), ('iso_512x32', 'iso'), ('diag_16x24', 'diag'),
                                      ('rough_4x16', 'rough')])
def test_g3_trajectories(tag, kind):
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    g = load('g3_trajectories')
    X0, V0 = g[tag + '_X0'], g[tag + '_V0']
    eps, L = float(g[tag + '_hp'][0]), int(g[tag + '_hp'][1])
    fake = dict(kind=kind, par_sigma=1.3, par_scale1=100, par_scale2=4,
                par_conditioning=10 ** np.linspace(-2, 0, X0.shape[0]))
    s = MarkovJumpHMC(distribution=product_distribution(fake, X0), epsilon=eps, beta=0.3, num_leapfrog_steps=L,
                      Vinit=V0, seed=3)
    st = s.state
    assert close(st.EX[0], g[tag + '_EX0']) and close(st.EV[0], g[tag + '_EV0']) and close(st.dEdX, g[tag + '_g0'])
    n = X0.shape[1]
    exps = np.stack([np.full(n, 1e-300), np.full(n, 1e300), np.full(n, 1e300)])
    s.sampling_iteration(replay=[(np.zeros_like(X0), exps)])
    assert np.all(s._dev.read(8) == 0)
    st = s.state
    assert close(st.X, g[tag + '_L_X']) and close(st.V, g[tag + '_L_V'])
    assert close(st.EX[0], g[tag + '_L_EX']) and close(st.EV[0], g[tag + '_L_EV'])
    assert close(st.dEdX, g[tag + '_L_g'])
    if kind == 'diag':
        assert bits_equal(st.X, g[tag + '_L_X']) and bits_equal(st.V, g[tag + '_L_V'])
    # the cached inverse-L state of an L-mover is the pre-move state: H_flf == H(state before)
    assert close(st.H_flf[0], g[tag + '_EX0'] + g[tag + '_EV0'])
    assert st.cache_active.all()


# ---------------------------------------------------------------------------------------------
# G4 / G6: full sampling_iteration replays, incl. the halve-epsilon retry
# ---------------------------------------------------------------------------------------------
G4 = ['g4_iso_2x100_a', 'g4_iso_2x100_b', 'g4_diag_16x64', 'g4_iso_512x32', 'g4_rough_4x48', 'g4_mm_3x40',
      'g4_iso_33x17', 'g6_retry_a_iso_4x32', 'g6_retry_b_iso_4x32']


@pytest.mark.parametrize('name', G4)
def test_g4_g6_sampling_iteration_replay(name, capsys):
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    g = load(name)
    d = product_distribution(g, g['Xinit'])
    s = MarkovJumpHMC(distribution=d, epsilon=float(g['eps']), beta=float(g['beta']), num_leapfrog_steps=int(g['L']),
                      Vinit=g['normals'][0], seed=5, resample=False)
    assert s.p_r == float(g['p_r'])
    feed = [(g['normals'][1 + a], np.nan_to_num(g['exps'][a], nan=1.0)) for a in range(len(g['exps']))]
    exact_state = str(g['kind']) == 'diag'
    for t in range(int(g['T']) + 1):
        if t:
            s.sampling_iteration(replay=feed)
            assert np.array_equal(s._dev.read(8), g['trans'][t - 1]), (name, t)
            assert close(s.dwelling_times, g['dwell'][t]), (name, t)
        st = s.state
        assert close(st.X, g['X'][t]) and close(st.V, g['V'][t]), (name, t)
        assert close(st.EX[0], g['EX'][t]) and close(st.EV[0], g['EV'][t]), (name, t)
        if exact_state:
            assert bits_equal(st.X, g['X'][t]) and bits_equal(st.V, g['V'][t]), (name, t)
        assert np.array_equal(st.cache_active, g['cache'][t]), (name, t)
        assert [s.l_count, s.f_count, s.r_count, s.fl_count] == list(g['counts'][t]), (name, t)
        assert [d.E_count, d.dEdX_count] == list(g['evals'][t]), (name, t)
        assert (s.epsilon, s.num_leapfrog_steps) == (g['hp'][t][0], int(g['hp'][t][1]))
        assert len(g['exps']) - len(feed) == int(g['attempts_done'][t])
        if str(g['kind']) in ('rough', 'mm'):
            # transcendental forces: ulp-level differences in sin/exp grow chaotically over many
            # iterations, so every iteration starts from the reference's state (per-iteration parity
            # from identical inputs).  Writing X/V clears the FLF cache; put it back.
            hflf = s._dev.read(5)
            assert np.array_equal(~np.isnan(hflf), g['cache'][t])
            s._dev.write(0, g['X'][t])
            s._dev.write(1, g['V'][t])
            s._dev.write(5, hflf)
    if name.startswith('g6'):
        assert 'doubling back' in capsys.readouterr().out


def test_g5_sample_with_resampling(monkeypatch):
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    g = load('g5_sample_2x100')
    d = product_distribution(g, g['Xinit'])
    s = MarkovJumpHMC(distribution=d, epsilon=float(g['eps']), beta=float(g['beta']), num_leapfrog_steps=int(g['L']),
                      Vinit=g['normals'][0], seed=5)
    feed = [(g['normals'][1 + a], np.nan_to_num(g['exps'][a], nan=1.0)) for a in range(len(g['exps']))]
    monkeypatch.setattr(np.random, 'random', lambda n: g['resample_u'].copy())
    out = s.sample(int(g['n_samples']), replay=feed)
    assert out.shape == (2, 1000)
    # the oracle run gives the reference's resampling indices (integer bookkeeping: exact)
    o = orc.MarkovJumpHMC(orc.IsoGaussian(1.0), g['Xinit'], epsilon=float(g['eps']), beta=float(g['beta']),
                          num_leapfrog_steps=int(g['L']), V0=g['normals'][0],
                          rng=orc.ReplayRNG(normals=list(g['normals'][1:]), exps=list(g['exps']), uniforms=[g['resample_u']]))
    ref = o.sample(int(g['n_samples']))
    assert bits_equal(ref, g['samples'])
    assert np.array_equal(s._last_resample_idx, o.last_pick)
    assert bits_equal(out, g['samples'])          # sigma = 1: the force is exact, so samples are bit-identical
    assert [s.l_count, s.f_count, s.r_count, s.fl_count] == list(g['counts'])
    assert [d.E_count, d.dEdX_count] == list(g['evals'])


# ---------------------------------------------------------------------------------------------
# production RNG (Philox keyed by global particle id) against the oracle's restatement of it
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('kind,D,N,eps,L,beta', [('iso', 2, 100, 0.3, 5, 0.3), ('iso', 512, 96, 0.05, 10, 0.1),
                                                 ('diag', 33, 70, 0.5, 4, 0.4), ('funnel', 32, 300, 0.05, 15, 0.1),
                                                 ('rough', 7, 130, 0.5, 6, 0.2), ('mm', 3, 64, 0.3, 6, 0.4)])
def test_philox_mode_matches_oracle(kind, D, N, eps, L, beta):
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc import distributions as Dm
    rng = np.random.RandomState(42)
    X0 = rng.randn(D, N)
    if kind == 'iso':
        base, kw, en = Dm.TestGaussian, dict(sigma=1.0), orc.IsoGaussian(1.0)
    elif kind == 'diag':
        base, kw, en = Dm.Gaussian, dict(log_conditioning=2), orc.DiagGaussian(D, 2)
    elif kind == 'funnel':
        base, kw, en = Dm.Funnel, dict(scale=3.0), orc.FunnelNeal(3.0)
        X0[0] *= 3.0
        X0[1:] *= np.exp(X0[0] / 2)
    elif kind == 'rough':
        base, kw, en = Dm.RoughWell, dict(), orc.RoughWell(100, 4)
        X0 *= 100
    else:
        base, kw, en = Dm.MultimodalGaussian, dict(separation=3), orc.MultimodalGaussian(D, 3)

    class Fixed(base):
        def init_X(self):
            self.Xinit = X0

    d = Fixed(ndims=D, nbatch=N, **kw)
    seed = 0xC0FFEE1234
    s = MarkovJumpHMC(distribution=d, epsilon=eps, beta=beta, num_leapfrog_steps=L, seed=seed, resample=False)
    o = orc.MarkovJumpHMC(en, X0, epsilon=eps, beta=beta, num_leapfrog_steps=L, resample=False,
                          rng=orc.PhiloxRNG(seed, np.arange(N)))
    assert close(s.state.V, o.state.V)                    # tick-0 momentum
    for t in range(6):
        s.sampling_iteration()
        o.sampling_iteration()
        assert np.array_equal(s._dev.read(8), o.last_transition), (kind, t)
        assert close(s.state.X, o.state.X) and close(s.state.V, o.state.V), (kind, t)
        assert close(s.state.EX, o.state.EX) and close(s.state.EV, o.state.EV), (kind, t)
        assert close(s.dwelling_times, o.dwelling_times), (kind, t)
        assert np.array_equal(s.state.cache_active, o.state.shadow_ok)
        assert (s.l_count, s.f_count, s.r_count) == (o.l_count, o.f_count, o.r_count)
        assert (d.E_count, d.dEdX_count) == (en.E_count, en.dEdX_count)
        if kind == 'rough':
            # sin/cos forces at |x| ~ 100 amplify ulp-level differences ~30x per iteration (measured);
            # restart every iteration from the oracle's state: parity per iteration, identical inputs
            hflf = s._dev.read(5)
            s._dev.write(0, o.state.X)
            s._dev.write(1, o.state.V)
            s._dev.write(5, hflf)


def test_batched_iterations_equal_single_steps():
    """mjhmc_iterate(n) with no host round trip == n calls of sampling_iteration()."""
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import TestGaussian
    np.random.seed(3)
    X0 = np.random.randn(64, 500)

    class Fixed(TestGaussian):
        def init_X(self):
            self.Xinit = X0
    a = MarkovJumpHMC(distribution=Fixed(64, 500), epsilon=0.2, beta=0.2, num_leapfrog_steps=7, seed=9, resample=False)
    b = MarkovJumpHMC(distribution=Fixed(64, 500), epsilon=0.2, beta=0.2, num_leapfrog_steps=7, seed=9, resample=False)
    xa = a.sample(12, preserve_order=True)
    xb = np.stack([(b.sampling_iteration(), b.state.X)[1] for _ in range(12)], axis=-1)
    assert xa.shape == (64, 500, 12) and bits_equal(xa, xb)
    assert (a.l_count, a.f_count, a.r_count) == (b.l_count, b.f_count, b.r_count)
    assert a.l_count + a.f_count + a.r_count == 12 * 500
    assert bits_equal(a.dwelling_times, b.dwelling_times)
    flat = Fixed(64, 500)
    c = MarkovJumpHMC(distribution=flat, epsilon=0.2, beta=0.2, num_leapfrog_steps=7, seed=9, resample=False)
    xc = c.sample(12)
    assert xc.shape == (64, 6000) and bits_equal(xc, np.concatenate([xa[:, :, t] for t in range(12)], axis=1))


def test_sharding_is_invisible():
    """Columns split over two samplers (as over two GPUs) with global particle ids give exactly the
    columns of the unsplit run: the RNG is keyed by global id, particles never interact."""
    from mjhmc_amd import engine, _lib
    rng = np.random.RandomState(8)
    D, N = 32, 1000
    X0 = rng.randn(D, N)
    X0[0] *= 3
    ctx = engine.context(0)
    en = engine.DeviceEnergy(ctx, _lib.E_FUNNEL_NEAL, D, [3.0])
    whole = engine.DeviceSampler(en, X0, seed=77)
    parts = [engine.DeviceSampler(en, X0[:, :400], seed=77, first_particle_id=0),
             engine.DeviceSampler(en, X0[:, 400:], seed=77, first_particle_id=400)]
    for smp in [whole] + parts:
        smp.set_hparams(0.05, 15, 0.05, 1.0)
        stats, done = smp.iterate(5)
        assert done == 5
    X = np.concatenate([p.read(_lib.F_X) for p in parts], axis=1)
    V = np.concatenate([p.read(_lib.F_V) for p in parts], axis=1)
    assert bits_equal(X, whole.read(_lib.F_X)) and bits_equal(V, whole.read(_lib.F_V))
    assert np.array_equal(np.concatenate([p.read(_lib.F_TRANS) for p in parts]), whole.read(_lib.F_TRANS))


# ---------------------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties + a column subset against the oracle
# ---------------------------------------------------------------------------------------------
def test_full_size_c2_properties_and_column_subset():
    """configs[1]: isotropic Gaussian, ndims=512, nparticles=100000, L=10, float64."""
    from mjhmc_amd import engine, _lib
    D, N, L, eps, beta = 512, 100000, 10, 0.05, 0.1
    rng = np.random.RandomState(0)
    X0 = rng.randn(D, N)
    ctx = engine.context(0)
    en = engine.DeviceEnergy(ctx, _lib.E_ISO_GAUSS, D, [1.0])
    s = engine.DeviceSampler(en, X0, seed=2024)
    p_r = -np.log(1 - beta) * 0.5
    s.set_hparams(eps, L, p_r, 1.0)
    H0 = s.read(_lib.F_EX) + s.read(_lib.F_EV)
    T = 4
    stats, done = s.iterate(T)
    assert done == T
    n_cold_expected = N
    for st in stats:
        assert st.l + st.f + st.r == N and st.nonfinite == 0
        assert st.n_cold == n_cold_expected                 # cold == not an L-mover last time
        assert st.E_evals == N + st.n_cold and st.dEdX_evals == L * (N + st.n_cold)
        n_cold_expected = N - st.l
    trans, cache = s.read(_lib.F_TRANS), s.read(_lib.F_CACHE)
    assert np.array_equal(cache == 1, trans == 0)           # exactly the L-movers hold a cached FLF state
    X, V = s.read(_lib.F_X), s.read(_lib.F_V)
    EX, EV = s.read(_lib.F_EX), s.read(_lib.F_EV)
    assert close(EX, np.sum(X ** 2, axis=0) / 2.) and close(EV, np.sum(V ** 2, axis=0) / 2.)
    # leapfrog at eps = 0.05 conserves H to O(eps^2); R-movers get fresh momentum, skip them
    # column subset through the oracle with the same counter RNG (keyed by global particle id)
    cols = np.sort(rng.choice(N, size=48, replace=False))
    o = orc.MarkovJumpHMC(orc.IsoGaussian(1.0), X0[:, cols], epsilon=eps, beta=beta, num_leapfrog_steps=L,
                          resample=False, rng=orc.PhiloxRNG(2024, cols))
    for _ in range(T):
        o.sampling_iteration()
    assert np.array_equal(trans[cols], o.last_transition)
    assert close(X[:, cols], o.state.X) and close(V[:, cols], o.state.V)
    assert close(EX[cols], o.state.EX[0]) and close(s.read(_lib.F_DWELL)[cols], o.dwelling_times)
    assert np.isfinite(H0).all()


# ---------------------------------------------------------------------------------------------
# G7 / G8: discrete-time control samplers and ContinuousTimeHMC
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name', ['g7_control_iso_2x100', 'g7_hmc_diag_8x32', 'g7_base_iso_3x50'])
def test_g7_control_samplers_replay(name):
    from mjhmc_amd.samplers import markov_jump_hmc as M
    g = load(name)
    d = product_distribution(g, g['Xinit'])
    cls = getattr(M, str(g['cls']))
    s = cls(distribution=d, epsilon=float(g['eps']), beta=float(g['beta_in']), num_leapfrog_steps=int(g['L']),
            Vinit=g['normals'][0], seed=5)
    assert (s.beta, s.p_r, s.p_flip) == (float(g['beta']), float(g['p_r']), float(g['p_flip']))
    N = g['Xinit'].shape[1]
    used = 1
    exact_state = str(g['kind']) == 'diag' and float(g['beta']) == 1.0
    for t in range(int(g['T']) + 1):
        if t:
            fired = g['u_r'][t - 1] < float(g['p_r'])
            noise = g['normals'][used] if fired else np.zeros_like(g['Xinit'])
            used += int(fired)
            unif = np.concatenate([g['u_acc'][t - 1], g['u_flip'][t - 1], [g['u_r'][t - 1]]])
            s.sampling_iteration(replay=[(noise, unif)])
            assert used == int(g['normals_done'][t])
        st = s.state
        assert close(st.X, g['X'][t]) and close(st.V, g['V'][t]), (name, t)
        assert close(st.EX[0], g['EX'][t]) and close(st.EV[0], g['EV'][t]), (name, t)
        if exact_state:
            assert bits_equal(st.X, g['X'][t]) and bits_equal(st.V, g['V'][t])
        assert [s.l_count, s.f_count, s.r_count, s.fl_count] == list(g['counts'][t]), (name, t)
        assert [d.E_count, d.dEdX_count] == list(g['evals'][t]), (name, t)


def test_g8_continuous_time_hmc_replay():
    from mjhmc_amd.samplers.markov_jump_hmc import ContinuousTimeHMC
    g = load('g8_cthmc_diag_6x40')
    d = product_distribution(g, g['Xinit'])
    s = ContinuousTimeHMC(distribution=d, epsilon=float(g['eps']), beta=float(g['beta']),
                          num_leapfrog_steps=int(g['L']), Vinit=g['normals'][1], seed=5, resample=False)
    feed = [(g['normals'][2 + a], np.nan_to_num(g['exps'][a], nan=1.0)) for a in range(len(g['exps']))]
    for t in range(int(g['T']) + 1):
        if t:
            s.sampling_iteration(replay=feed)
            # device codes 0 = FL, 1 = F, 2 = R ; fixture rows follow min_idx([f, fl, r])
            want = np.array([1, 0, 2], dtype=np.uint8)[g['trans'][t - 1]]
            assert np.array_equal(s._dev.read(8), want), t
            assert close(s.dwelling_times, g['dwell'][t]), t
        st = s.state
        assert bits_equal(st.X, g['X'][t]) and bits_equal(st.V, g['V'][t]), t
        assert close(st.EX[0], g['EX'][t]) and close(st.EV[0], g['EV'][t]), t
        assert [s.l_count, s.f_count, s.r_count, s.fl_count] == list(g['counts'][t]), t
        assert [d.E_count, d.dEdX_count] == list(g['evals'][t]), t


@pytest.mark.parametrize('cls_name,kind,D,N,eps,L,beta', [('ControlHMC', 'iso', 16, 200, 0.3, 5, 0.5),
                                                          ('HMC', 'diag', 9, 150, 0.4, 4, 0.4),
                                                          ('HMCBase', 'iso', 3, 64, 0.3, 6, 0.6),
                                                          ('ContinuousTimeHMC', 'iso', 64, 90, 0.2, 5, 0.3)])
def test_philox_mode_control_and_ct(cls_name, kind, D, N, eps, L, beta):
    from mjhmc_amd.samplers import markov_jump_hmc as M
    from mjhmc_amd.misc import distributions as Dm
    rng = np.random.RandomState(4)
    X0 = rng.randn(D, N)
    base, kw, en = ((Dm.TestGaussian, dict(sigma=1.0), orc.IsoGaussian(1.0)) if kind == 'iso'
                    else (Dm.Gaussian, dict(log_conditioning=2), orc.DiagGaussian(D, 2)))

    class Fixed(base):
        def init_X(self):
            self.Xinit = X0

    d = Fixed(ndims=D, nbatch=N, **kw)
    seed = 31337
    extra = dict(resample=False) if cls_name == 'ContinuousTimeHMC' else {}
    s = getattr(M, cls_name)(distribution=d, epsilon=eps, beta=beta, num_leapfrog_steps=L, seed=seed, **extra)
    o = getattr(orc, cls_name)(en, X0, epsilon=eps, beta=beta, num_leapfrog_steps=L,
                               rng=orc.PhiloxRNG(seed, np.arange(N)), **extra)
    assert (s.beta, s.p_r, s.p_flip) == (o.beta, o.p_r, o.p_flip)
    for t in range(8):
        s.sampling_iteration()
        o.sampling_iteration()
        assert close(s.state.X, o.state.X) and close(s.state.V, o.state.V), (cls_name, t)
        assert close(s.state.EX, o.state.EX) and close(s.state.EV, o.state.EV), (cls_name, t)
        assert (s.l_count, s.f_count, s.r_count, s.fl_count) == (o.l_count, o.f_count, o.r_count, o.fl_count)
        assert (d.E_count, d.dEdX_count) == (en.E_count, en.dEdX_count)
    assert s.r_count > 0 or cls_name == 'ContinuousTimeHMC'


# ---------------------------------------------------------------------------------------------
# configs[0]: the README example, and the reference's statistical acceptance tests
# ---------------------------------------------------------------------------------------------
def test_readme_example_runs_unchanged():
    """README.md:12-37 with only the import lines switched (BASELINE.json configs[0])."""
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import LambdaDistribution

    def E(X, sigma=1.):
        return np.sum(X ** 2, axis=0).reshape((1, -1)) / 2. / sigma ** 2

    def dEdX(X, sigma=1.):
        return X / sigma ** 2

    np.random.seed(0)
    Xinit = np.random.randn(2, 100)
    anonymous_gaussian = LambdaDistribution(energy_func=E, energy_grad_func=dEdX, init=Xinit, name='IsotropicGaussian')
    mjhmc = MarkovJumpHMC(distribution=anonymous_gaussian)
    X = mjhmc.sample(num_steps=10)
    assert X.shape == (2, 1000) and np.isfinite(X).all()
    assert mjhmc.l_count + mjhmc.f_count + mjhmc.r_count == 11 * 100
    assert mjhmc.r_count == 0                      # defaults give beta = 0.2**2000 = 0 -> p_r = 0
    assert anonymous_gaussian.E_count >= 100 * 12


# ---------------------------------------------------------------------------------------------
# LambdaDistribution beyond the isotropic Gaussian (README.md:27-36: the callables define the distribution)
# ---------------------------------------------------------------------------------------------
def _student_like(D):
    w = np.linspace(0.5, 3.0, D)

    def E(X):
        return np.sum(np.log(1.0 + X ** 2 / w[:, None]), axis=0).reshape((1, -1))

    def dEdX(X):
        return 2.0 * X / (w[:, None] + X ** 2)
    return w, E, dEdX


@pytest.mark.parametrize('cls_name,D,N', [('MarkovJumpHMC', 10, 90), ('MarkovJumpHMC', 300, 33), ('ControlHMC', 7, 64),
                                          ('ContinuousTimeHMC', 40, 50)])
def test_lambda_distribution_with_device_expressions(cls_name, D, N):
    """A separable energy stated as C expressions (compiled with hipRTC around the engine's kernels) next to the
    NumPy callables: the sampler must follow the oracle run on the CALLABLES, transitions exactly, state to 1e-10."""
    from mjhmc_amd.samplers import markov_jump_hmc as M
    from mjhmc_amd.misc.distributions import LambdaDistribution
    w, E, dEdX = _student_like(D)
    X0 = np.random.RandomState(D).randn(D, N) * 2.0
    d = LambdaDistribution(energy_func=E, energy_grad_func=dEdX, init=X0, name='student-like',
                           device_expr=("log(1.0 + x*x/p[d])", "2.0*x/(p[d] + x*x)"), device_params=w)
    assert close(d.E(X0)[0], E(X0)[0]) and close(d.dEdX(X0), dEdX(X0))
    kw = dict(epsilon=0.3, beta=0.4, num_leapfrog_steps=5)
    extra = dict(resample=False) if cls_name != 'ControlHMC' else {}
    s = getattr(M, cls_name)(distribution=d, seed=61, **kw, **extra)
    en = orc.LambdaEnergy(E, dEdX)
    o = getattr(orc, cls_name)(en, X0, rng=orc.PhiloxRNG(61, np.arange(N)), **kw, **extra)
    for t in range(8):
        s.sampling_iteration()
        o.sampling_iteration()
        if cls_name == 'MarkovJumpHMC':
            assert np.array_equal(s._dev.read(8), o.last_transition), t
            assert close(s.dwelling_times, o.dwelling_times), t
        assert close(s.state.X, o.state.X) and close(s.state.V, o.state.V), t
        assert close(s.state.EX, o.state.EX) and close(s.state.EV, o.state.EV), t
        assert (s.l_count, s.f_count, s.r_count, s.fl_count) == (o.l_count, o.f_count, o.r_count, o.fl_count), t
    out = s.sample(5)                                             # batched launches + the sample ring
    assert out.shape == (D, 5 * N) and np.isfinite(out).all()
    Z = s.state.copy().L()                                        # the snapshot operators go through the same energy
    o.state.X[:], o.state.V[:] = s.state.X, s.state.V
    o.state.refresh_EX(); o.state.refresh_EV(); o.state.refresh_grad()
    Zo = o.state.clone().L()
    assert close(Z.X, Zo.X) and close(Z.V, Zo.V) and close(Z.EX, Zo.EX)


FUNNEL_EXPR = dict(stats=["d == 0 ? x : 0.0", "d == 0 ? 0.0 : x*x"], energy="0.0",
                   energy0="S[0]*S[0]/(2*p[0]*p[0]) + 0.5*exp(-S[0])*S[1] + 0.5*(p[1]-1)*S[0]",
                   grad="d == 0 ? x/(p[0]*p[0]) - 0.5*exp(-x)*S[1] + 0.5*(p[1]-1) : x*exp(-S[0])")


@pytest.mark.parametrize('cls_name,D,N', [('MarkovJumpHMC', 32, 70), ('MarkovJumpHMC', 10, 200), ('ControlHMC', 32, 40)])
def test_lambda_distribution_with_coupled_expressions(cls_name, D, N):
    """Coordinates coupled through per-particle statistics S[k] (mjhmc_energy_create_expr_coupled): Neal's funnel
    written as user expressions must follow the oracle's FunnelNeal -- and the built-in FUNNEL_NEAL functor."""
    from mjhmc_amd.samplers import markov_jump_hmc as M
    from mjhmc_amd.misc.distributions import LambdaDistribution, Funnel
    rs = np.random.RandomState(D + N)
    x0 = 1.5 * rs.randn(N)
    X0 = np.vstack([x0, np.exp(x0 / 2) * rs.randn(D - 1, N)])
    en = orc.FunnelNeal(3.0)
    d = LambdaDistribution(energy_func=en.E_val, energy_grad_func=en.dEdX_val, init=X0, name='funnel as expressions',
                           device_expr=FUNNEL_EXPR, device_params=[3.0, float(D)])
    assert close(d.E(X0)[0], en.E_val(X0)[0]) and close(d.dEdX(X0), en.dEdX_val(X0))
    builtin = Funnel(ndims=D, nbatch=N, scale=3.0)
    assert close(d.E(X0)[0], builtin.E(X0)[0]) and close(d.dEdX(X0), builtin.dEdX(X0))
    kw = dict(epsilon=0.05, beta=0.3, num_leapfrog_steps=6)
    extra = dict(resample=False) if cls_name != 'ControlHMC' else {}
    s = getattr(M, cls_name)(distribution=d, seed=77, **kw, **extra)
    o = getattr(orc, cls_name)(en, X0, rng=orc.PhiloxRNG(77, np.arange(N)), **kw, **extra)
    for t in range(6):
        s.sampling_iteration()
        o.sampling_iteration()
        if cls_name == 'MarkovJumpHMC':
            assert np.array_equal(s._dev.read(8), o.last_transition), t
            assert close(s.dwelling_times, o.dwelling_times), t
        assert close(s.state.X, o.state.X) and close(s.state.V, o.state.V), t
        assert close(s.state.EX, o.state.EX) and close(s.state.EV, o.state.EV), t
        assert (s.l_count, s.f_count, s.r_count, s.fl_count) == (o.l_count, o.f_count, o.r_count, o.fl_count), t
    out = s.sample(4)
    assert out.shape == (D, 4 * N) and np.isfinite(out).all()


def test_lambda_distribution_checks_the_expressions_against_the_callables():
    from mjhmc_amd import _lib
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import LambdaDistribution
    w, E, dEdX = _student_like(6)
    X0 = np.random.RandomState(1).randn(6, 20)
    wrong = LambdaDistribution(energy_func=E, energy_grad_func=dEdX, init=X0,
                               device_expr=("log(1.0 + x*x/p[d])", "x/(p[d] + x*x)"), device_params=w)   # gradient off by 2
    with pytest.raises(ValueError):
        MarkovJumpHMC(distribution=wrong, epsilon=0.1, beta=0.1)
    broken = LambdaDistribution(init=X0, device_expr=("log(1.0 + x*x/q)", "x"))
    with pytest.raises(_lib.EngineError) as info:
        MarkovJumpHMC(distribution=broken, epsilon=0.1, beta=0.1)
    assert "undeclared identifier 'q'" in str(info.value)
    with pytest.raises(NotImplementedError):                                          # no callables, no expressions
        MarkovJumpHMC(distribution=LambdaDistribution(init=X0), epsilon=0.1, beta=0.1)


def _dense_quadratic(D, seed):
    """E(x) = x^T A x / 2 with a DENSE symmetric positive definite A: not separable, not expressible through a few
    per-particle statistics -- only the callables can evaluate it."""
    rs = np.random.RandomState(seed)
    Q = rs.randn(D, D)
    A = Q.dot(Q.T) / D + 0.5 * np.eye(D)

    def E(X):
        return 0.5 * np.sum(X * A.dot(X), axis=0).reshape((1, -1))

    def dEdX(X):
        return A.dot(X)
    return A, E, dEdX


@pytest.mark.parametrize('cls_name,D,N', [('MarkovJumpHMC', 6, 90), ('MarkovJumpHMC', 37, 70), ('ControlHMC', 9, 64),
                                          ('HMC', 5, 40), ('ContinuousTimeHMC', 12, 50)])
def test_lambda_distribution_with_opaque_callables(cls_name, D, N):
    """README.md:27-36: LambdaDistribution(energy_func, energy_grad_func, init) with callables the engine has no device
    form of (a dense quadratic form).  State, leapfrog arithmetic, jump decision, commit and counters stay on the
    device; the gradient comes from the Python callable once per leapfrog step (mjhmc_traj_*).  The run must follow
    the oracle on the same callables and the same Philox streams: transitions exact, state to 1e-10, the evaluation
    counters exact."""
    from mjhmc_amd import _lib
    from mjhmc_amd.samplers import markov_jump_hmc as M
    from mjhmc_amd.misc.distributions import LambdaDistribution
    A, E, dEdX = _dense_quadratic(D, D + N)
    X0 = np.random.RandomState(N).randn(D, N) * 1.5
    calls = {'E': 0, 'g': 0}

    def E_c(X):
        calls['E'] += 1
        return E(X)

    def g_c(X):
        calls['g'] += 1
        return dEdX(X)
    d = LambdaDistribution(energy_func=E_c, energy_grad_func=g_c, init=X0)            # two callables and nothing else
    assert d.device_energy()[0] == _lib.E_HOST
    kw = dict(epsilon=0.25, beta=0.4, num_leapfrog_steps=5)
    extra = dict(resample=False) if cls_name not in ('ControlHMC', 'HMC') else {}
    s = getattr(M, cls_name)(distribution=d, seed=19, **kw, **extra)
    en = orc.LambdaEnergy(E, dEdX)
    o = getattr(orc, cls_name)(en, X0, rng=orc.PhiloxRNG(19, np.arange(N)), **kw, **extra)
    assert close(s.state.V, o.state.V) and close(s.state.EX, o.state.EX) and close(s.state.dEdX, o.state.dEdX)
    for t in range(10):
        before = dict(calls)
        s.sampling_iteration()
        o.sampling_iteration()
        assert calls['g'] - before['g'] == 5 and calls['E'] - before['E'] == 1       # one call-back per leapfrog step
        if cls_name == 'MarkovJumpHMC':
            assert np.array_equal(s._dev.read(8), o.last_transition), t
            assert np.array_equal(s.state.cache_active, o.state.shadow_ok), t
        if cls_name in ('MarkovJumpHMC', 'ContinuousTimeHMC'):
            assert close(s.dwelling_times, o.dwelling_times), t
        assert close(s.state.X, o.state.X) and close(s.state.V, o.state.V), t
        assert close(s.state.EX, o.state.EX) and close(s.state.EV, o.state.EV) and close(s.state.dEdX, o.state.dEdX), t
        assert (s.l_count, s.f_count, s.r_count, s.fl_count) == (o.l_count, o.f_count, o.r_count, o.fl_count), t
        assert (d.E_count, d.dEdX_count) == (en.E_count, en.dEdX_count), t
    assert s.r_count + s.f_count + s.l_count + s.fl_count > 0
    # sample(): the device ring and, for the jump processes, the dwell-time resampling
    np.random.seed(3)
    out = s.sample(6)
    np.random.seed(3)
    want = o.sample(6)
    assert out.shape == want.shape and close(out, want)
    # HMCState assignment re-evaluates E and dE/dX of the new positions through the callables
    st = s.state.copy()
    st.X = st.X * 0.5
    s.state = st
    assert close(s.state.EX, E(st.X)) and close(s.state.dEdX, dEdX(st.X))


def test_opaque_callables_meet_a_non_finite_rate_like_the_reference():
    """markov_jump_hmc.py:376-389 with host-evaluated energies: a huge step makes exp(H0 - H1) overflow for some
    particle, the attempt is not committed, epsilon is halved / L doubled until it goes through -- as the oracle does."""
    import contextlib
    import io
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import LambdaDistribution
    D, N = 5, 64
    A, E, dEdX = _dense_quadratic(D, 1)
    X0 = np.random.RandomState(2).randn(D, N) * 400.0            # H ~ 1e5: a relative leapfrog error of 1e-2 exceeds log(DBL_MAX)
    d = LambdaDistribution(energy_func=E, energy_grad_func=dEdX, init=X0)
    kw = dict(epsilon=0.5, beta=0.3, num_leapfrog_steps=4)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        s = MarkovJumpHMC(distribution=d, seed=5, resample=False, **kw)
        o = orc.MarkovJumpHMC(orc.LambdaEnergy(E, dEdX), X0, rng=orc.PhiloxRNG(5, np.arange(N)), resample=False, **kw)
        for t in range(3):
            s.sampling_iteration()
            o.sampling_iteration()
            assert np.array_equal(s._dev.read(8), o.last_transition), t
            assert close(s.state.X, o.state.X) and close(s.state.V, o.state.V), t
    assert buf.getvalue().count('doubling back') == len(o.retry_depths) > 0
    assert (s.epsilon, s.num_leapfrog_steps) == (0.5, 4)
    assert (s.l_count, s.f_count, s.r_count) == (o.l_count, o.f_count, o.r_count)


def test_lambda_distribution_recognises_a_diagonal_gaussian():
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import LambdaDistribution
    j = 10 ** np.linspace(-2, 0, 12)
    X0 = np.random.RandomState(2).randn(12, 50)
    d = LambdaDistribution(energy_func=lambda X: 0.5 * np.sum(j[:, None] * X ** 2, axis=0).reshape((1, -1)),
                           energy_grad_func=lambda X: j[:, None] * X, init=X0)
    s = MarkovJumpHMC(distribution=d, epsilon=0.4, beta=0.3, num_leapfrog_steps=5, seed=9, resample=False)
    en = orc.DiagGaussian(12, 2)
    o = orc.MarkovJumpHMC(en, X0, epsilon=0.4, beta=0.3, num_leapfrog_steps=5, resample=False,
                          rng=orc.PhiloxRNG(9, np.arange(50)))
    for t in range(6):
        s.sampling_iteration()
        o.sampling_iteration()
        assert np.array_equal(s._dev.read(8), o.last_transition) and close(s.state.X, o.state.X), t


@pytest.mark.parametrize('cls_name', ['MarkovJumpHMC', 'ControlHMC', 'HMC', 'HMCBase'])
def test_statistical_1d_gaussian(cls_name):
    """mjhmc/tests/test_continuous_samplers.py:19-41 with usable hyper-parameters: mean, std within 0.05."""
    from mjhmc_amd.samplers import markov_jump_hmc as M
    from mjhmc_amd.misc.distributions import TestGaussian
    np.random.seed(1)
    extra = dict(resample=False) if cls_name == 'MarkovJumpHMC' else {}
    s = getattr(M, cls_name)(distribution=TestGaussian(ndims=1, nbatch=100), epsilon=0.5, beta=0.3,
                             num_leapfrog_steps=3, seed=11, **extra)
    s.burn_in()
    samples = s.sample(10000)
    assert samples.shape == (1, 100 * 10000)
    if cls_name == 'MarkovJumpHMC':
        s2 = M.MarkovJumpHMC(distribution=TestGaussian(ndims=1, nbatch=100), epsilon=0.5, beta=0.3,
                             num_leapfrog_steps=3, seed=12)          # resample=True: dwell-time weighted
        s2.burn_in()
        samples = s2.sample(2000)
    assert abs(np.mean(samples)) < 0.05, np.mean(samples)
    assert abs(np.std(samples) - 1) < 0.05, np.std(samples)


@pytest.mark.parametrize('cls_name', ['MarkovJumpHMC', 'ControlHMC'])
def test_statistical_ill_conditioned_gaussian(cls_name):
    """mjhmc/tests/test_continuous_samplers.py:43-63: 2-D Gaussian, conditioning 10, covariance error < 0.05
    (relative Frobenius norm here: the target covariance has an entry of 10)."""
    from mjhmc_amd.samplers import markov_jump_hmc as M
    from mjhmc_amd.misc.distributions import Gaussian
    np.random.seed(1)
    g = Gaussian(ndims=2, nbatch=200, log_conditioning=1)
    target = np.linalg.inv(g.J)
    s = getattr(M, cls_name)(distribution=g, epsilon=0.6, beta=0.2, num_leapfrog_steps=4, seed=21)
    s.burn_in()
    samples = s.sample(3000)
    err = np.linalg.norm(np.cov(samples) - target) / np.linalg.norm(target)
    assert err < 0.05, err


# ---------------------------------------------------------------------------------------------
# ProductOfT on the matrix cores (float32; parity unpinned by reference tests -> oracle in float64)
# ---------------------------------------------------------------------------------------------
def pot_weights(ndims, seed=2015):
    """init_weights of mjhmc/search/MJHMC_poe_36/mjhmc_objective.py:15-23 (+ I to keep W invertible)."""
    rs = np.random.RandomState(seed)
    sp_var = rs.rand(ndims, ndims)
    w_sp = rs.randn(ndims, ndims)
    w_sp[sp_var > 0.05] = 0
    lognu = np.log(rs.rand(ndims) * 2 + 2.1)
    return w_sp + np.eye(ndims), lognu


@pytest.mark.parametrize('ndims,n', [(36, 25), (100, 40), (128, 33), (200, 64), (256, 31), (512, 70)])
def test_pot_energy_and_gradient(ndims, n):
    from mjhmc_amd.misc.distributions import ProductOfT
    W, lognu = pot_weights(ndims)
    b = 0.1 * np.random.RandomState(1).randn(ndims)
    d = ProductOfT(ndims=ndims, nbasis=ndims, nbatch=n, lognu=lognu, W=W, b=b, state_dtype='float32')
    o = orc.ProductOfT(W, lognu=lognu, b=b, force_dtype=np.float64)
    X = np.random.RandomState(2).randn(ndims, n) * 1.5
    E, G = d.E(X), d.dEdX(X)
    Eo, Go = o.E_val(X), o.dEdX_val(X)
    assert E.shape == (1, n) and G.shape == (ndims, n)
    assert np.allclose(E, Eo, rtol=2e-5, atol=2e-5 * np.abs(Eo).max())
    assert np.allclose(G, Go, rtol=0, atol=3e-5 * np.abs(Go).max())


@pytest.mark.parametrize('ndims,N,eps,L,beta', [(36, 50, 0.1, 6, 0.3), (200, 70, 0.08, 6, 0.3), (512, 96, 0.05, 8, 0.2)])
def test_pot_iterations_vs_oracle(ndims, N, eps, L, beta):
    """Per-iteration parity from identical inputs (float32 device vs float64 oracle with the end points stored in
    float32): every transition equal or a proven near tie, state and energies within float32 tolerance."""
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import ProductOfT
    from tests.helpers import check_iteration, resync
    W, lognu = pot_weights(ndims)
    X0 = np.random.RandomState(3).randn(ndims, N)

    class Fixed(ProductOfT):
        def init_X(self):
            self.Xinit = X0

    d = Fixed(ndims=ndims, nbasis=ndims, nbatch=N, lognu=lognu, W=W, state_dtype='float32')
    en = orc.ProductOfT(W, lognu=lognu, force_dtype=np.float64)
    seed = 99
    s = MarkovJumpHMC(distribution=d, epsilon=eps, beta=beta, num_leapfrog_steps=L, seed=seed, resample=False)
    o = orc.MarkovJumpHMC(en, X0, epsilon=eps, beta=beta, num_leapfrog_steps=L, resample=False,
                          rng=orc.PhiloxRNG(seed, np.arange(N)),
                          state_rounding=lambda a: a.astype(np.float32).astype(np.float64))
    assert np.allclose(s.state.V, o.state.V, atol=1e-6)
    assert np.allclose(s.state.EX, o.state.EX, rtol=2e-5) and np.allclose(s.state.dEdX, o.state.dEdX, atol=1e-4)
    resync(s, o)                       # start both from the float32-rounded state so the inputs are identical
    ties = 0
    for t in range(5):
        ties += check_iteration(s, o, delta_rel=2e-5, x_tol=2e-5, e_rtol=2e-5, tag='pot %d it %d' % (ndims, t))
        assert s.l_count + s.f_count + s.r_count == (t + 1) * N
        resync(s, o)                   # next iteration starts from the device state on both sides
    assert ties <= 2, ties             # near ties are rare: O(energy error) of the particles


# ---------------------------------------------------------------------------------------------
# SparseImageCode: bf16 state / bf16 MFMA operands / fp32 accumulate (parity unpinned -> oracle in float64)
# ---------------------------------------------------------------------------------------------
def sic_problem(seed=0):
    """Synthetic dictionary: column-normalised B (256, 1024), patch y = B a0 + noise."""
    rs = np.random.RandomState(seed)
    B = rs.randn(256, 1024)
    B /= np.linalg.norm(B, axis=0, keepdims=True)
    a0 = rs.randn(1024) * (rs.rand(1024) < 0.05)
    y = B.dot(a0) + 0.1 * rs.randn(256)
    return B, y.reshape(256, 1), a0


def to_bf16(a):
    """round-to-nearest-even float64 -> bfloat16 -> float64 (what the device stores)"""
    u = np.asarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


@pytest.mark.parametrize('cauchy', [True, False])
def test_sic_energy_and_gradient(cauchy):
    from mjhmc_amd.misc.distributions import SparseImageCode
    B, y, a0 = sic_problem()
    n = 70
    rs = np.random.RandomState(1)
    X = to_bf16(a0[:, None] + 0.3 * rs.randn(1024, n))           # representable in the state dtype
    d = SparseImageCode(n_patches=1, n_batches=n, cauchy=cauchy, n_basis=1024, basis=B, imgs=y, init=X, state_dtype='bfloat16')
    o = orc.SparseImageCode(to_bf16(B), y.T, lmbda=0.01, cauchy=cauchy)   # the dictionary is an MFMA operand: bf16
    E, G = d.E(X), d.dEdX(X)
    Eo, Go = o.E_val(X), o.dEdX_val(X)
    assert E.shape == (1, n) and G.shape == (1024, n)
    assert np.allclose(E, Eo, rtol=2e-3), np.abs(E / Eo - 1).max()
    # gradient: the residual is rounded to bf16 before the second GEMM
    assert np.abs(G - Go).max() < 2e-2 * np.abs(Go).max()


# Energy tolerance of the bf16 kernel against the mixed-precision oracle, relative to max|H|: the two sides agree to
# float32 accumulation error EXCEPT where a value sits on a bf16 rounding boundary and is stored one bf16 ulp apart
# (|v| ~ 3 -> ulp 2^-6 -> 0.05 in EV at |H| ~ 200 = 2.3e-4 per such element; measured up to 2.2e-4 over 5 iterations).
SIC_E_TOL = 5e-4


def test_sic_iterations_vs_oracle():
    """bf16 kernel against the mixed-precision restatement of the oracle (bf16 operands of both matrix products, end
    points stored in bf16, everything else float64): transitions equal or proven near ties.  epsilon is a power of
    two so that rounding the scaled residual equals scaling the rounded residual."""
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import SparseImageCode
    from tests.helpers import check_iteration, resync
    B, y, a0 = sic_problem()
    N, eps, L, beta, seed = 96, 0.0625, 10, 0.2, 5
    X0 = to_bf16(a0[:, None] + 0.2 * np.random.RandomState(2).randn(1024, N))
    d = SparseImageCode(n_patches=1, n_batches=N, cauchy=True, n_basis=1024, basis=B, imgs=y, init=X0, state_dtype='bfloat16')
    en = orc.SparseImageCode(B, y.T, lmbda=0.01, cauchy=True, operand_rounding=to_bf16)
    s = MarkovJumpHMC(distribution=d, epsilon=eps, beta=beta, num_leapfrog_steps=L, seed=seed, resample=False)
    o = orc.MarkovJumpHMC(en, X0, epsilon=eps, beta=beta, num_leapfrog_steps=L, resample=False,
                          rng=orc.PhiloxRNG(seed, np.arange(N)), state_rounding=to_bf16)
    V0 = s.state.V
    assert np.abs(V0 - o.state.V).max() < 2e-2                    # bf16-rounded tick-0 momentum
    assert np.array_equal(V0, to_bf16(V0))
    resync(s, o)
    assert np.allclose(s.state.EX, o.state.EX, rtol=1e-4) and np.allclose(s.state.EV, o.state.EV, rtol=1e-5)
    ties = 0
    for t in range(5):
        ties += check_iteration(s, o, delta_rel=SIC_E_TOL, x_tol=1.0 / 128, e_rtol=SIC_E_TOL, tag='sic it %d' % t)
        assert s.l_count + s.f_count + s.r_count == (t + 1) * N
        Xd, Vd = s.state.X, s.state.V
        assert np.array_equal(Xd, to_bf16(Xd)) and np.array_equal(Vd, to_bf16(Vd))
        resync(s, o)
    assert ties <= 2, ties


# ---------------------------------------------------------------------------------------------
# the reference's main caller: autocor.generate_samples (batched on the device)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cls_name', ['MarkovJumpHMC', 'ControlHMC'])
def test_generate_samples_batched_equals_per_step_loop(cls_name):
    """Batched driver == the reference's literal loop (smp.sample(1) + counter reads per step,
    mjhmc/misc/autocor.py:242-248) on a twin sampler: samples bit-identical, counter traces exact."""
    from mjhmc_amd.samplers import markov_jump_hmc as M
    from mjhmc_amd.misc.autocor import generate_samples
    from mjhmc_amd.misc.distributions import Gaussian
    np.random.seed(5)
    X0 = np.random.randn(10, 60)

    class Fixed(Gaussian):
        def init_X(self):
            self.Xinit = X0
    kw = dict(epsilon=0.4, beta=0.3, num_leapfrog_steps=5, seed=77)
    if cls_name == 'MarkovJumpHMC':
        kw['resample'] = False
    cls = getattr(M, cls_name)
    T = 40
    d1 = Fixed(ndims=10, nbatch=60, log_conditioning=2)
    samples, e_evals, grad_evals = generate_samples(cls, d1, num_steps=T, **kw)
    d2 = Fixed(ndims=10, nbatch=60, log_conditioning=2)
    smp = cls(distribution=d2, **kw)
    d2.E_count = d2.dEdX_count = 0
    ref_s, ref_e, ref_g = np.zeros((10, 60, T)), np.zeros(T), np.zeros(T)
    for t in range(T):
        ref_s[:, :, t] = smp.sample(1)
        ref_g[t] = d2.dEdX_count / 60.0
        ref_e[t] = d2.E_count / 60.0
    assert samples.shape == (10, 60, T) and bits_equal(samples, ref_s)
    assert np.array_equal(e_evals, ref_e) and np.array_equal(grad_evals, ref_g)
    # gradient-budget form (objective.py passes num_grad_steps): stops at the first step reaching the budget
    d3 = Fixed(ndims=10, nbatch=60, log_conditioning=2)
    s3, e3, g3 = generate_samples(cls, d3, num_grad_steps=100, **kw)
    k = int(np.nonzero(ref_g >= 100)[0][0]) + 1
    assert s3.shape[2] == k and np.array_equal(g3, ref_g[:k]) and bits_equal(s3, ref_s[:, :, :k])


# ---------------------------------------------------------------------------------------------
# edge shapes: a single particle (experiments/spectral.py runs nbatch == 1), slot boundaries, D = 1
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('D,N', [(1, 1), (3, 1), (1, 63), (2, 64), (5, 65), (513, 3), (1024, 2), (40, 129)])
def test_edge_shapes_match_oracle(D, N):
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import TestGaussian
    X0 = np.random.RandomState(D * 1000 + N).randn(D, N)

    class Fixed(TestGaussian):
        def init_X(self):
            self.Xinit = X0

    d = Fixed(ndims=D, nbatch=N, sigma=1.0)
    s = MarkovJumpHMC(distribution=d, epsilon=0.2, beta=0.4, num_leapfrog_steps=4, seed=123, resample=False)
    en = orc.IsoGaussian(1.0)
    o = orc.MarkovJumpHMC(en, X0, epsilon=0.2, beta=0.4, num_leapfrog_steps=4, resample=False,
                          rng=orc.PhiloxRNG(123, np.arange(N)))
    for t in range(6):
        s.sampling_iteration()
        o.sampling_iteration()
        assert np.array_equal(s._dev.read(8), o.last_transition), (D, N, t)
        assert close(s.state.X, o.state.X) and close(s.state.V, o.state.V), (D, N, t)
        assert s.state.H().shape == (1, N)
    assert (s.l_count, s.f_count, s.r_count) == (o.l_count, o.f_count, o.r_count)
    assert (d.E_count, d.dEdX_count) == (en.E_count, en.dEdX_count)
    out = s.sample(3, preserve_order=True)                  # one fused launch of three iterations
    for t in range(3):
        o.sampling_iteration()
    assert out.shape == (D, N, 3) and close(out[:, :, 2], o.state.X)
    assert np.array_equal(s._dev.read(8), o.last_transition)
    assert (s.l_count, s.f_count, s.r_count) == (o.l_count, o.f_count, o.r_count)


def test_unsupported_shapes_fail_loudly():
    """What is left: input the REFERENCE rejects too (tf_distributions.py:219 asserts the dictionary shape), and dtypes
    that name no arithmetic of an energy.  Every ndims / dtype combination of the other energies runs (below)."""
    from mjhmc_amd import _lib, engine
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import TestGaussian, SparseImageCode
    B = np.random.RandomState(0).randn(64, 128)
    d = SparseImageCode(n_patches=1, n_batches=4, n_basis=128, basis=B, imgs=np.zeros((64, 1)))
    with pytest.raises(_lib.EngineError):                      # only the 256 x 1024 / 256 x 512 dictionary shapes exist
        MarkovJumpHMC(distribution=d, epsilon=0.1, beta=0.1)
    ctx = engine.context(0)
    en = engine.DeviceEnergy(ctx, _lib.E_ISO_GAUSS, 8, [1.0])
    with pytest.raises(_lib.EngineError):                      # bfloat16 state is SparseImageCode's arithmetic only
        engine.DeviceSampler(en, np.zeros((8, 4)), seed=1, dtype='bfloat16')


@pytest.mark.parametrize('kind,D,N', [('iso', 2500, 40), ('funnel', 2100, 33)])
def test_float32_state_beyond_the_register_kernels(kind, D, N):
    """float32 state for rows wider than the float32 register kernels hold (64 lanes x 32 elements): the multi-pass path on
    float64 storage, the state rounded to float32 wherever it is written.  Against the oracle with the same rounding
    (end points of every iteration), from identical inputs: transitions exact, state at float32 resolution, and every
    stored value IS a float32."""
    from mjhmc_amd import engine, _lib
    rs = np.random.RandomState(D)
    ctx = engine.context(0)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)           # noqa: E731
    if kind == 'iso':
        X0, en_o = rs.randn(D, N), orc.IsoGaussian(1.3)
        en = engine.DeviceEnergy(ctx, _lib.E_ISO_GAUSS, D, [1.3])
        eps, L = 0.1, 5
    else:
        x0 = 0.5 * rs.randn(N)
        X0, en_o = np.vstack([x0, np.exp(x0 / 2) * rs.randn(D - 1, N)]), orc.FunnelNeal(3.0)
        en = engine.DeviceEnergy(ctx, _lib.E_FUNNEL_NEAL, D, [3.0])
        eps, L = 0.02, 4
    s = engine.DeviceSampler(en, X0, seed=7, dtype='float32')
    p_r = -np.log(1 - 0.3) * 0.5
    s.set_hparams(eps, L, p_r, 1.0)
    o = orc.MarkovJumpHMC(en_o, f32(X0), epsilon=eps, beta=0.3, num_leapfrog_steps=L, resample=False,
                          rng=orc.PhiloxRNG(7, np.arange(N)), state_rounding=f32)
    X, V = s.read(_lib.F_X), s.read(_lib.F_V)
    assert np.array_equal(X, f32(X0)) and np.array_equal(V, f32(V)) and np.allclose(V, o.state.V, atol=1e-6)
    for t in range(4):
        o.state.X[:], o.state.V[:] = s.read(_lib.F_X), s.read(_lib.F_V)          # identical inputs every iteration
        o.state.refresh_EX(); o.state.refresh_EV(); o.state.refresh_grad()
        hf = s.read(_lib.F_HFLF)
        o.state.shadow_ok[:] = ~np.isnan(hf)
        o.state.shadow.EX[0, :] = np.nan_to_num(hf)
        o.state.shadow.EV[0, :] = 0.0
        st, done = s.iterate(1)
        o.sampling_iteration()
        assert done == 1 and np.array_equal(s.read(_lib.F_TRANS), o.last_transition), t
        X, V = s.read(_lib.F_X), s.read(_lib.F_V)
        assert np.array_equal(X, f32(X)) and np.array_equal(V, f32(V))           # float32 values
        assert np.allclose(X, o.state.X, rtol=0, atol=2e-7 * max(1.0, np.abs(o.state.X).max()))
        assert np.allclose(V, o.state.V, rtol=0, atol=2e-7 * max(1.0, np.abs(o.state.V).max()))
        assert np.allclose(s.read(_lib.F_EX), o.state.EX[0], rtol=1e-6) and np.allclose(s.read(_lib.F_EV), o.state.EV[0], rtol=1e-6)
    out, _ = en.eval(X0[:, :8], dtype='float32')
    assert np.allclose(out, np.asarray(en_o.E_val(f32(X0[:, :8]))).reshape(-1), rtol=1e-10)
    s.close()


@pytest.mark.parametrize('cls_name,kind,D,N', [('MarkovJumpHMC', 'iso', 1100, 70), ('MarkovJumpHMC', 'diag', 5000, 33),
                                               ('MarkovJumpHMC', 'funnel', 1500, 40), ('ControlHMC', 'rough', 1030, 50),
                                               ('ContinuousTimeHMC', 'mm', 1300, 20)])
def test_more_dims_than_the_register_kernels_hold(cls_name, kind, D, N):
    """hmc_state.py:86-100 are plain array operations: the reference takes any ndims.  Beyond 1024 float64 dims (64 lanes x
    16 elements) the engine switches to its multi-pass path -- state in HBM between the substeps, one wavefront per row
    (csrc/host_energy.hip: multipass_iterate) -- and must follow the oracle exactly as the register kernels do."""
    from mjhmc_amd.samplers import markov_jump_hmc as M
    from mjhmc_amd.misc import distributions as Dm
    rs = np.random.RandomState(D + N)
    if kind == 'iso':
        X0, en, kw = rs.randn(D, N), orc.IsoGaussian(1.3), dict(epsilon=0.1, beta=0.3, num_leapfrog_steps=5)
        mk = lambda: Dm.TestGaussian(ndims=D, nbatch=N, sigma=1.3)                       # noqa: E731
    elif kind == 'diag':
        X0, en, kw = rs.randn(D, N), orc.DiagGaussian(D, 2), dict(epsilon=0.1, beta=0.3, num_leapfrog_steps=4)
        mk = lambda: Dm.Gaussian(ndims=D, nbatch=N, log_conditioning=2)                 # noqa: E731
    elif kind == 'funnel':
        x0 = 0.5 * rs.randn(N)
        X0 = np.vstack([x0, np.exp(x0 / 2) * rs.randn(D - 1, N)])
        en, kw = orc.FunnelNeal(3.0), dict(epsilon=0.02, beta=0.3, num_leapfrog_steps=4)
        mk = lambda: Dm.Funnel(ndims=D, nbatch=N, scale=3.0)                            # noqa: E731
    elif kind == 'rough':
        X0, en, kw = 3.0 * rs.randn(D, N), orc.RoughWell(100.0, 4.0), dict(epsilon=0.3, beta=0.3, num_leapfrog_steps=3)
        mk = lambda: Dm.RoughWell(ndims=D, nbatch=N, scale1=100, scale2=4)              # noqa: E731
    else:
        X0, en, kw = 0.3 * rs.randn(D, N), orc.MultimodalGaussian(D, 1), dict(epsilon=0.02, beta=0.3, num_leapfrog_steps=3)
        mk = lambda: Dm.MultimodalGaussian(ndims=D, nbatch=N, separation=1)          # noqa: E731

    def fixed():
        d = mk()
        d.Xinit = X0
        d.gen_init_X = lambda: setattr(d, 'Xinit', X0)
        d.init_X = lambda: setattr(d, 'Xinit', X0)
        return d
    d = fixed()
    assert close(np.ravel(d.E(X0)), np.ravel(en.E_val(X0))) and close(d.dEdX(X0), en.dEdX_val(X0))     # mjhmc_eval on wide rows
    extra = dict(resample=False) if cls_name != 'ControlHMC' else {}
    s = getattr(M, cls_name)(distribution=d, seed=23, **kw, **extra)
    o = getattr(orc, cls_name)(en, X0, rng=orc.PhiloxRNG(23, np.arange(N)), **kw, **extra)
    for t in range(5):
        s.sampling_iteration()
        o.sampling_iteration()
        if cls_name == 'MarkovJumpHMC':
            assert np.array_equal(s._dev.read(8), o.last_transition), t
            assert close(s.dwelling_times, o.dwelling_times), t
        assert close(s.state.X, o.state.X) and close(s.state.V, o.state.V), t
        assert close(s.state.EX, o.state.EX) and close(s.state.EV, o.state.EV) and close(s.state.dEdX, o.state.dEdX), t
        assert (s.l_count, s.f_count, s.r_count, s.fl_count) == (o.l_count, o.f_count, o.r_count, o.fl_count), t
        assert (d.E_count, d.dEdX_count) == (en.E_count, en.dEdX_count), t
    out = s.sample(3, preserve_order=True)                           # batched call + the sample ring
    for t in range(3):
        o.sampling_iteration()
    assert out.shape == (D, N, 3) and close(out[:, :, 2], o.state.X)
    Z = s.state.copy().L()                                           # mjhmc_leapfrog on wide rows
    Zo = o.state.clone().L()
    assert close(Z.X, Zo.X) and close(Z.V, Zo.V) and close(Z.EX, Zo.EX) and close(Z.EV, Zo.EV)


# ---------------------------------------------------------------------------------------------
# fair-initialisation generator (gen_mj_init.py) -- online variance known answer, pickle format
# ---------------------------------------------------------------------------------------------
def test_online_variance_known_answer_and_init_cache(tmp_path):
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import Gaussian
    from mjhmc_amd.misc import gen_mj_init as G
    np.random.seed(2)
    d = Gaussian(ndims=6, nbatch=50, log_conditioning=1)
    a = MarkovJumpHMC(distribution=d, epsilon=0.5, beta=0.3, num_leapfrog_steps=4, seed=8, resample=False)
    var, _ = G.online_variance(a, d, var_steps=300, block=64)
    d2 = Gaussian(ndims=6, nbatch=50, log_conditioning=1)
    d2.Xinit = d.Xinit
    d2.init_X = lambda: None
    b = MarkovJumpHMC(distribution=d2, epsilon=0.5, beta=0.3, num_leapfrog_steps=4, seed=8, resample=False)
    vals = b.sample(300, preserve_order=True)                      # the same 300 states
    assert abs(var - np.var(vals.ravel(), ddof=1)) < 1e-12 * np.var(vals.ravel())
    # generator end to end at toy step counts; pickle layout of the reference
    d3 = Gaussian(ndims=4, nbatch=40, log_conditioning=1)
    path = G.cache_initialization(d3, str(tmp_path), burn_in_steps=400, var_steps=200, seed=3)
    mj, emc_var, true_var, ctl = G.load_initialization(d3, str(tmp_path))
    assert mj.shape == (4, 40) and ctl.shape == (4, 40) and np.isfinite(mj).all() and np.isfinite(ctl).all()
    assert emc_var > 0 and true_var > 0
    assert os.path.basename(path).startswith('Gaussian_') and G.stable_digest(d3) in path
