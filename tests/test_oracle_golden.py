"""CPU: the NumPy oracle reproduces the reference bit-for-bit on every golden vector.

The fixtures were captured from the imported reference by oracle/capture_golden.py.
"""
import numpy as np
import pytest

from oracle import mjhmc_oracle as orc
from tests.helpers import load, oracle_energy, bits_equal

np.seterr(all='ignore')


def test_g1_min_idx():
    g = load('g1_min_idx')                                    # mjhmc/tests/test_utils.py:15-52
    (i0, i1), _ = orc.first_minimum([g['a'].reshape(1, -1), g['b'].reshape(1, -1)])
    assert np.array_equal(i0, g['two_0']) and np.array_equal(i1, g['two_1'])
    assert np.array_equal(i0, np.arange(100)[g['a'] < g['b']])
    (j0, j1, j2), _ = orc.first_minimum([g[k].reshape(1, -1) for k in 'cde'])
    assert np.array_equal(j0, g['three_0']) and np.array_equal(j1, g['three_1']) and np.array_equal(j2, g['three_2'])
    (t0, t1, t2), _ = orc.first_minimum([g['t'], g['u'], g['v']])
    assert np.array_equal(t0, g['ties_0']) and np.array_equal(t1, g['ties_1']) and np.array_equal(t2, g['ties_2'])


@pytest.mark.parametrize('tag,kind,par', [
    ('iso_2x100', 'iso', dict(sigma=1.3)), ('iso_512x64', 'iso', dict(sigma=1.3)),
    ('diag_10x33', 'diag', dict(ndims=10, log_conditioning=6)), ('rough_5x40', 'rough', dict(scale1=100, scale2=4)),
    ('mm_3x20', 'mm', dict(ndims=3, separation=3))])
def test_g2_energy_evaluations(tag, kind, par):
    g = load('g2_energies')
    cls = {'iso': orc.IsoGaussian, 'diag': orc.DiagGaussian, 'rough': orc.RoughWell, 'mm': orc.MultimodalGaussian}[kind]
    e = cls(**par)
    X = g[tag + '_X']
    assert bits_equal(np.asarray(e.E_val(X)).reshape(-1), g[tag + '_E'])
    assert bits_equal(e.dEdX_val(X), g[tag + '_g'])
    idx = np.arange(0, X.shape[1], 2)
    assert bits_equal(np.asarray(e.E_val(X[:, idx])).reshape(-1), g[tag + '_Eg'])
    assert bits_equal(e.dEdX_val(X[:, idx]), g[tag + '_gg'])


@pytest.mark.parametrize('tag,kind', [('iso_2x100', 'iso'), ('iso_512x32', 'iso'), ('diag_16x24', 'diag'), ('rough_4x16', 'rough')])
def test_g3_trajectories(tag, kind):
    g = load('g3_trajectories')
    X0, V0 = g[tag + '_X0'], g[tag + '_V0']
    eps, L = g[tag + '_hp']
    if kind == 'iso':
        en = orc.IsoGaussian(sigma=1.3)
    elif kind == 'diag':
        en = orc.DiagGaussian(ndims=X0.shape[0], log_conditioning=2)
    else:
        en = orc.RoughWell(100, 4)
    s = orc.MarkovJumpHMC(en, X0, epsilon=float(eps), beta=0.3, num_leapfrog_steps=int(L), V0=V0, rng=orc.ReplayRNG())
    assert bits_equal(s.state.EX[0], g[tag + '_EX0']) and bits_equal(s.state.EV[0], g[tag + '_EV0'])
    assert bits_equal(s.state.dEdX, g[tag + '_g0'])
    for nm, z in (('L', s.state.clone().L()), ('FLF', s.state.clone().FLF())):
        for f, arr in (('X', z.X), ('V', z.V), ('EX', z.EX[0]), ('EV', z.EV[0]), ('g', z.dEdX)):
            assert bits_equal(arr, g['%s_%s_%s' % (tag, nm, f)]), (tag, nm, f)


G4 = ['g4_iso_2x100_a', 'g4_iso_2x100_b', 'g4_diag_16x64', 'g4_iso_512x32', 'g4_rough_4x48', 'g4_mm_3x40',
      'g4_iso_33x17', 'g6_retry_a_iso_4x32', 'g6_retry_b_iso_4x32']


@pytest.mark.parametrize('name', G4)
def test_g4_g6_sampling_iteration_replay(name):
    g = load(name)
    en = oracle_energy(g)
    rng = orc.ReplayRNG(normals=list(g['normals'][1:]), exps=list(g['exps']))
    s = orc.MarkovJumpHMC(en, g['Xinit'], epsilon=float(g['eps']), beta=float(g['beta']),
                          num_leapfrog_steps=int(g['L']), V0=g['normals'][0], rng=rng, resample=False)
    assert s.p_r == float(g['p_r'])
    T = int(g['T'])
    for t in range(T + 1):
        if t:
            s.sampling_iteration()
            assert np.array_equal(s.last_transition, g['trans'][t - 1]), (name, t)
            assert bits_equal(s.dwelling_times, g['dwell'][t]), (name, t)
        st = s.state
        for f, arr in (('X', st.X), ('V', st.V), ('EX', st.EX[0]), ('EV', st.EV[0])):
            assert bits_equal(arr, g[f][t]), (name, t, f)
        if 'dEdX' in g.files:
            assert bits_equal(st.dEdX, g['dEdX'][t])
        assert np.array_equal(st.shadow_ok, g['cache'][t])
        assert [s.l_count, s.f_count, s.r_count, s.fl_count] == list(g['counts'][t])
        assert [en.E_count, en.dEdX_count] == list(g['evals'][t])
        assert rng.attempt == int(g['attempts_done'][t])
        assert (s.epsilon, s.num_leapfrog_steps) == (g['hp'][t][0], int(g['hp'][t][1]))
    if name.startswith('g6'):
        assert rng.attempt > T                                  # the retry path really ran


def test_g5_sample_with_resampling():
    g = load('g5_sample_2x100')
    en = orc.IsoGaussian(sigma=1.0)
    rng = orc.ReplayRNG(normals=list(g['normals'][1:]), exps=list(g['exps']), uniforms=[g['resample_u']])
    s = orc.MarkovJumpHMC(en, g['Xinit'], epsilon=float(g['eps']), beta=float(g['beta']),
                          num_leapfrog_steps=int(g['L']), V0=g['normals'][0], rng=rng)
    out = s.sample(int(g['n_samples']))
    assert out.shape == (2, 1000)
    assert bits_equal(out, g['samples'])
    assert [s.l_count, s.f_count, s.r_count, s.fl_count] == list(g['counts'])
    assert [en.E_count, en.dEdX_count] == list(g['evals'])


@pytest.mark.parametrize('name', ['g7_control_iso_2x100', 'g7_hmc_diag_8x32', 'g7_base_iso_3x50'])
def test_g7_control_samplers(name):
    g = load(name)
    en = oracle_energy(g)
    uni = []
    for t in range(int(g['T'])):
        uni += [g['u_acc'][t], g['u_flip'][t], g['u_r'][t]]
    rng = orc.ReplayRNG(normals=list(g['normals'][1:]), uniforms=uni)
    cls = getattr(orc, str(g['cls']))
    s = cls(en, g['Xinit'], epsilon=float(g['eps']), beta=float(g['beta_in']), num_leapfrog_steps=int(g['L']),
            V0=g['normals'][0], rng=rng)
    assert (s.beta, s.p_r, s.p_flip) == (float(g['beta']), float(g['p_r']), float(g['p_flip']))
    for t in range(int(g['T']) + 1):
        if t:
            s.sampling_iteration()
        st = s.state
        for f, arr in (('X', st.X), ('V', st.V), ('EX', st.EX[0]), ('EV', st.EV[0]), ('dEdX', st.dEdX)):
            assert bits_equal(arr, g[f][t]), (name, t, f)
        assert [s.l_count, s.f_count, s.r_count, s.fl_count] == list(g['counts'][t])
        assert [en.E_count, en.dEdX_count] == list(g['evals'][t])
        assert rng.n_normals_used == int(g['normals_done'][t]) - 1


def test_global_rng_mode_equals_replay():
    """Seeding np.random and letting the oracle draw in the reference's order gives the same chain
    as replaying what the reference drew (pins the draw ORDER, utils.py:37-42 / hmc_state.py:125)."""
    g = load('g4_diag_16x64')
    en = oracle_energy(g)
    np.random.seed(103)
    X0 = np.random.randn(16, 64)
    assert bits_equal(X0, g['Xinit'])
    s = orc.MarkovJumpHMC(en, X0, epsilon=float(g['eps']), beta=float(g['beta']), num_leapfrog_steps=int(g['L']),
                          resample=False)
    for t in range(1, int(g['T']) + 1):
        s.sampling_iteration()
        assert bits_equal(s.state.X, g['X'][t]) and bits_equal(s.state.V, g['V'][t])


def test_g8_continuous_time_hmc_replay():
    """ContinuousTimeHMC (F / FL / R clocks, markov_jump_hmc.py:251-290).  The reference builds the state
    twice for this class (HMCBase.__init__ then ContinuousTimeHMC.__init__), so V0 is the 2nd randn block."""
    g = load('g8_cthmc_diag_6x40')
    en = oracle_energy(g)
    rng = orc.ReplayRNG(normals=list(g['normals'][2:]), exps=list(g['exps']))
    s = orc.ContinuousTimeHMC(en, g['Xinit'], epsilon=float(g['eps']), beta=float(g['beta']),
                              num_leapfrog_steps=int(g['L']), V0=g['normals'][1], rng=rng, resample=False)
    for t in range(int(g['T']) + 1):
        if t:
            s.sampling_iteration()
            assert bits_equal(s.dwelling_times, g['dwell'][t])
        st = s.state
        for f, arr in (('X', st.X), ('V', st.V), ('EX', st.EX[0]), ('EV', st.EV[0]), ('dEdX', st.dEdX)):
            assert bits_equal(arr, g[f][t]), (t, f)
        assert [s.l_count, s.f_count, s.r_count, s.fl_count] == list(g['counts'][t])
        assert [en.E_count, en.dEdX_count] == list(g['evals'][t])


def test_autocor_oracle_against_its_definition():
    """fft_autocor (autocor.py:37-49) is the normalised circular lag sum; slow_autocorrelation (:177-211)
    the lag-product mean.  The reference module is not importable (py2 prints, mklfft): pin the
    restatement on the explicit definitions, on a real sample block of the golden G4 run."""
    from oracle import autocor_oracle as ac
    g = load("g4_diag_16x64")
    X = np.stack([g['X'][t] for t in range(g['X'].shape[0])], axis=-1)      # (D, N, T)
    circ = ac.circular_lag_sums(X)
    np.testing.assert_allclose(ac.fft_autocor(X), circ / circ[0], rtol=0, atol=1e-12)
    lin = ac.linear_lag_sums(X)
    T = X.shape[2]
    means = lin / (X.shape[0] * X.shape[1] * (T - np.arange(T)))
    slow, _, _ = ac.slow_autocorrelation(X, None, None, half_window=False)
    np.testing.assert_allclose(slow, means[:T - 1] / means[0], rtol=1e-13)
    slow_h, _, _ = ac.slow_autocorrelation(X, None, None, half_window=True)
    assert slow_h.shape == (T // 2 - 1,)
    np.testing.assert_allclose(slow_h, means[:T // 2 - 1] / means[0], rtol=1e-13)
    e = np.arange(T, dtype=float)
    brute, e2, _ = ac.autocorrelation(X, e, e, half_window=False, brute_force=True)
    assert brute.shape == (T - 1, 1) and e2.shape == (T - 1,)
    np.testing.assert_allclose(brute[:, 0], means[:T - 1] / means[0], rtol=1e-13)


# ---------------------------------------------------------------------------------------------
# G9: the reference's generate_samples (autocor.py:213-261) and the autocorrelation of its output
# (:37-49, :177-211), captured by executing those functions from the reference's own file
# ---------------------------------------------------------------------------------------------
def g9_replay_feed(g):
    """per-iteration (normals, other numbers) of a G9 run, in the form sampling_iteration(replay=...) takes"""
    if str(g['cls']) == 'MarkovJumpHMC':
        return [(g['normals'][1 + t], np.nan_to_num(g['exps'][t], nan=1.0)) for t in range(int(g['T']))]
    feed, used = [], 1
    for t in range(int(g['T'])):
        fired = g['u_r'][t] < float(g['p_r'])
        noise = g['normals'][used] if fired else np.zeros_like(g['Xinit'])
        used += int(fired)
        feed.append((noise, np.concatenate([g['u_acc'][t], g['u_flip'][t], [g['u_r'][t]]])))
    return feed


@pytest.mark.parametrize('name', ['g9_generate_mjhmc_diag_6x40', 'g9_generate_control_iso_3x50'])
def test_g9_generate_samples_and_autocorrelation(name):
    from oracle import autocor_oracle as ac
    g = load(name)
    en = oracle_energy(g)
    T, N = int(g['T']), g['Xinit'].shape[1]
    kw = dict(epsilon=float(g['eps']), beta=float(g['beta_in']), num_leapfrog_steps=int(g['L']), V0=g['normals'][0])
    if str(g['cls']) == 'MarkovJumpHMC':
        s = orc.MarkovJumpHMC(en, g['Xinit'], resample=False,
                              rng=orc.ReplayRNG(normals=list(g['normals'][1:]), exps=list(g['exps'])), **kw)
    else:
        unif = []
        for t in range(T):
            unif += [g['u_acc'][t], g['u_flip'][t], g['u_r'][t]]
        s = orc.ControlHMC(en, g['Xinit'], rng=orc.ReplayRNG(normals=list(g['normals'][1:]), uniforms=unif), **kw)
    en.E_count = en.dEdX_count = 0                         # distribution.reset() (autocor.py:241)
    samples = np.zeros_like(g['samples'])
    e_evals, grad_evals = np.zeros(T), np.zeros(T)
    for t in range(T):                                     # the loop of autocor.py:242-248
        samples[:, :, t] = s.sample(1)
        grad_evals[t] = en.dEdX_count / float(N)
        e_evals[t] = en.E_count / float(N)
    assert bits_equal(samples, g['samples'])
    assert np.array_equal(e_evals, g['e_evals']) and np.array_equal(grad_evals, g['grad_evals'])
    np.testing.assert_allclose(ac.fft_autocor(samples), g['fft_autocor'], rtol=0, atol=1e-13)
    np.testing.assert_allclose(ac.slow_autocorrelation(samples, e_evals, grad_evals)[0], g['slow_autocor'], rtol=1e-13)
    auto, e2, g2 = ac.autocorrelation(samples, e_evals, grad_evals, half_window=False)
    np.testing.assert_allclose(auto, g['fft_autocor'], rtol=0, atol=1e-13)
