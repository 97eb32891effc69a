"""Shared test plumbing: build oracle objects from golden fixtures."""
import os

import numpy as np

from oracle import mjhmc_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def oracle_energy(g, ndims=None):
    kind = str(g['kind'])
    if kind == 'iso':
        return orc.IsoGaussian(sigma=float(g['par_sigma']))
    if kind == 'diag':
        e = orc.DiagGaussian(ndims=len(g['par_conditioning']), log_conditioning=2)
        e.conditioning = g['par_conditioning']
        e.J = np.diag(e.conditioning)
        return e
    if kind == 'rough':
        return orc.RoughWell(scale1=int(g['par_scale1']), scale2=int(g['par_scale2']))
    if kind == 'mm':
        return orc.MultimodalGaussian(ndims=ndims or g['Xinit'].shape[0], separation=int(g['par_separation']))
    raise KeyError(kind)


def bits_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(np.all((a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))))


# ---------------------------------------------------------------------------------------------
# model recipes shared by the dense-energy fixtures (oracle/capture_dense_fixtures.py) and their tests
# ---------------------------------------------------------------------------------------------
def ref_init_weights(ndims, nbasis):
    """init_weights of mjhmc/search/MJHMC_poe_36/mjhmc_objective.py:15-23 (seed 2015), as the reference writes it."""
    rs = np.random.RandomState(2015)
    sp_var = rs.rand(ndims, nbasis)
    w_sp = rs.randn(ndims, nbasis)
    w_sp[sp_var > 0.05] = 0
    lognu = np.log(rs.rand(nbasis,) * 2 + 2.1)
    return w_sp, lognu


def sic_problem(seed=0, n_patches=1, img=256, n_coeffs=1024):
    """Synthetic sparse-coding problem (the reference's distr_data/dump_1024.pkl is not in its
    checkout): column-normalised dictionary B (img, n_coeffs), patches y_p = B a0_p + 0.1 noise with 5 %-sparse
    a0_p.  Returns B, imgs (img, n_patches) as the reference's data['data'] is laid out, a0 (n_patches * n_coeffs,)."""
    rs = np.random.RandomState(seed)
    B = rs.randn(img, n_coeffs)
    B /= np.linalg.norm(B, axis=0, keepdims=True)
    a0 = rs.randn(n_coeffs) * (rs.rand(n_coeffs) < 0.05)
    y = B.dot(a0) + 0.1 * rs.randn(img)
    cols, codes = [y], [a0]
    for _ in range(1, n_patches):
        a = rs.randn(n_coeffs) * (rs.rand(n_coeffs) < 0.05)
        cols.append(B.dot(a) + 0.1 * rs.randn(img))
        codes.append(a)
    return B, np.stack(cols, axis=1), np.concatenate(codes)


def to_bf16(a):
    """round-to-nearest-even float64 -> bfloat16 -> float64 (what the device stores)"""
    u = np.asarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


# ---------------------------------------------------------------------------------------------
# reduced-precision kernels (float32 ProductOfT, bf16 SparseImageCode): a transition that differs from the float64
# oracle must be a NEAR TIE -- the oracle itself changes its mind when the energy differences move by no more than
# the kernel's energy error -- and everything else must be equal.
# ---------------------------------------------------------------------------------------------
def oracle_proposal_energies(o):
    """(H0, HL, Hflf, unit exponentials (3, N)) the oracle's NEXT MarkovJumpHMC.sampling_iteration will use
    (markov_jump_hmc.py:356-368); evaluation counters are put back."""
    en = o.energy
    counts = (en.E_count, en.dEdX_count)
    H0 = o.state.H()[0].copy()
    HL = o.state.clone().L().H()[0].copy()
    Hflf = o.state.clone().FLF().H()[0].copy()
    en.E_count, en.dEdX_count = counts
    if hasattr(o.rng, 'stream'):          # counter RNG: the draws of the coming tick
        exps = np.asarray(o.rng.stream.unit_exponentials(o.rng.tick), dtype=np.float64)
    else:                                 # recorded numbers (orc.ReplayRNG): the coming attempt's block
        exps = np.asarray(o.rng._exps[o.rng.attempt], dtype=np.float64)
    return H0, HL, Hflf, exps


def _argmin_lfr(dHL, dHflf, p_r, exps):
    with np.errstate(all='ignore'):
        l = np.exp(dHL) ** .5
        flf = np.exp(dHflf) ** .5
        f = flf - np.minimum(flf, l)
        r = np.full_like(l, p_r)
        draws = []
        for rate, e in ((l, exps[0]), (f, exps[1]), (r, exps[2])):
            draws.append(np.where(rate == 0, np.inf, (1. / rate) * e))
    return np.argmin(np.stack(draws), axis=0)


def explainable_transitions(H0, HL, Hflf, p_r, exps, delta, device_trans, grid=5):
    """bool (N,): the device's transition is what the oracle's own decision rule yields for SOME perturbation of the
    two energy differences H0 - HL and H0 - Hflf by at most ``delta`` (scalar or (N,)) each."""
    ok = np.zeros(H0.shape[0], dtype=bool)
    steps = np.linspace(-1.0, 1.0, grid)
    for a in steps:
        for b in steps:
            ok |= _argmin_lfr(H0 - HL + a * delta, H0 - Hflf + b * delta, p_r, exps) == device_trans
    return ok


def resync(s, o, cols=None):
    """next iteration starts from the device state on both sides (reduced-precision states drift apart otherwise)"""
    sel = slice(None) if cols is None else cols
    Xd, Vd = s.state.X[:, sel], s.state.V[:, sel]
    hflf = s._dev.read(5)[sel]
    o.state.X[:], o.state.V[:] = Xd, Vd
    o.state.refresh_EX(); o.state.refresh_EV(); o.state.refresh_grad()
    o.state.shadow_ok[:] = ~np.isnan(hflf)
    o.state.shadow.EX[0, :] = np.nan_to_num(hflf)
    o.state.shadow.EV[0, :] = 0.0


def tie_ceiling(n):
    """How many of n compared particles may take another transition than the oracle's as PROVEN near ties before the
    comparison stops meaning anything: 2 of 48, 4 % of a big batch (measured: 0-2 per thousand)."""
    return max(2, int(0.04 * n))


def check_iteration(s, o, delta_rel, x_tol, e_rtol, tag='', cols=None, max_ties=None):
    """One MarkovJumpHMC.sampling_iteration on the device sampler ``s`` and on the oracle ``o`` (which may hold only the
    columns ``cols`` of the batch) from identical inputs.  Every transition must be the oracle's, or be a provable
    near tie at an energy error of ``delta_rel`` * max|H| -- and at most ``max_ties`` (default tie_ceiling) of them may
    be; state, energies, cache flags and dwelling times of the agreeing particles are compared.  Returns the number of
    near ties."""
    sel = slice(None) if cols is None else cols
    H0, HL, Hflf, exps = oracle_proposal_energies(o)
    s.sampling_iteration()
    o.sampling_iteration()
    tr, tro = s._dev.read(8)[sel], o.last_transition
    scale = max(1.0, float(np.abs(H0).max()))
    ok = explainable_transitions(H0, HL, Hflf, o.p_r, exps, delta_rel * scale, tr)
    diff = tr != tro
    if not ok.all():                      # say how far off the unexplained particles are before failing
        need = {}
        for mult in (2, 4, 8, 16, 64, 256, 4096):
            okm = explainable_transitions(H0, HL, Hflf, o.p_r, exps, mult * delta_rel * scale, tr)
            for i in np.nonzero(~ok & okm)[0]:
                need.setdefault(int(i), mult)
        raise AssertionError((tag, 'transitions not explained by an energy error of %g (max|H| %g): particle -> '
                                   'multiple of it that would: %s; never: %s'
                              % (delta_rel * scale, scale, need, [int(i) for i in np.nonzero(~ok)[0] if int(i) not in need])))
    limit = tie_ceiling(tr.shape[0]) if max_ties is None else max_ties
    assert int(diff.sum()) <= limit, (tag, 'explained near ties are no longer rare: %d of %d (ceiling %d)' % (diff.sum(), tr.shape[0], limit))
    same = ~diff
    Xd, Vd = s.state.X[:, sel], s.state.V[:, sel]
    xs = max(1.0, float(np.abs(o.state.X).max()))
    assert np.abs(Xd[:, same] - o.state.X[:, same]).max() <= x_tol * xs, (tag, 'X')
    assert np.abs(Vd[:, same] - o.state.V[:, same]).max() <= x_tol * max(1.0, float(np.abs(o.state.V).max())), (tag, 'V')
    eEX = np.abs(s.state.EX[0, sel][same] - o.state.EX[0, same]).max() / scale
    eEV = np.abs(s.state.EV[0, sel][same] - o.state.EV[0, same]).max() / scale
    assert eEX <= e_rtol and eEV <= e_rtol, (tag, 'EX, EV errors / max|H|', eEX, eEV, 'allowed', e_rtol)
    assert np.array_equal(s.state.cache_active[sel][same], o.state.shadow_ok[same]), (tag, 'cache flags')
    keep = same & np.isfinite(o.dwelling_times) & (tr != 1)
    # dwell = e / sqrt(exp(dH)): an energy error d moves it by d/2 relative (the F clock, flf - min(flf, l), is
    # ill-conditioned near flf == l and is covered by the transition check instead)
    assert np.allclose(s.dwelling_times[sel][keep], o.dwelling_times[keep], rtol=max(1e-3, 4 * delta_rel * scale)), (tag, 'dwell')
    return int(diff.sum())


def check_control_iteration(s, o, delta_rel, x_tol, e_rtol, tag='', max_ties=None):
    """One HMCBase / HMC / ControlHMC sampling_iteration (markov_jump_hmc.py:116-148) on the device sampler and on the
    oracle from identical inputs.  Flips and the batch-wide refresh involve no energies and must be equal; an
    accept decision may differ only where the acceptance test `u < exp(H0 - H1)` is within the kernel's energy
    error of a tie.  Returns the number of such near ties."""
    en = o.energy
    counts = (en.E_count, en.dEdX_count)
    H0 = o.state.H()[0].copy()
    HL = o.state.clone().L().H()[0].copy()
    en.E_count, en.dEdX_count = counts
    uacc = np.asarray(o.rng.stream.accept_uniforms(o.rng.tick), dtype=np.float64)
    n = H0.shape[0]
    r_before = o.r_count
    s.sampling_iteration()
    o.sampling_iteration()
    tr = s._dev.read(8)
    acc_o = np.isin(np.arange(n), o.last_fl_idx)
    flip_o = np.isin(np.arange(n), o.last_flip_idx)
    assert np.array_equal(((tr >> 1) & 1).astype(bool), flip_o), (tag, 'flips')
    diff = (tr & 1).astype(bool) != acc_o
    scale = max(1.0, float(np.abs(H0).max()))
    with np.errstate(all='ignore'):
        margin = np.abs((H0 - HL) - np.log(uacc))
    assert (margin[diff] <= delta_rel * scale).all(), (tag, 'accept decisions off by more than the energy error',
                                                       margin[diff], delta_rel * scale)
    assert (s.r_count > 0) == (o.r_count > 0) and (o.r_count - r_before) in (0, n), (tag, 'refresh gate')
    limit = tie_ceiling(n) if max_ties is None else max_ties
    assert int(diff.sum()) <= limit, (tag, 'explained near ties are no longer rare: %d of %d (ceiling %d)' % (diff.sum(), n, limit))
    same = ~diff
    xs = max(1.0, float(np.abs(o.state.X).max()))
    assert np.abs(s.state.X[:, same] - o.state.X[:, same]).max() <= x_tol * xs, (tag, 'X')
    assert np.abs(s.state.V[:, same] - o.state.V[:, same]).max() <= x_tol * max(1.0, float(np.abs(o.state.V).max())), (tag, 'V')
    eEX = np.abs(s.state.EX[0, same] - o.state.EX[0, same]).max() / scale
    eEV = np.abs(s.state.EV[0, same] - o.state.EV[0, same]).max() / scale
    assert eEX <= e_rtol and eEV <= e_rtol, (tag, 'EX, EV errors / max|H|', eEX, eEV, 'allowed', e_rtol)
    return int(diff.sum())


_HOOKS_CTX = {}


def hooks_context(device=0):
    """An engine.Context on libmjhmc_hip_test.so -- the SAME sources built with -DMJHMC_TEST_HOOKS (csrc/Makefile:
    test_hooks): only there do the environment A/B switches of the launch strategies (MJHMC_NO_FUSE, MJHMC_NO_COMPACT,
    MJHMC_NO_SPLIT, ...), the failure-placing hook MJHMC_DEBUG_POISON and the single-GPU gather hooks exist.  The A/B
    tests run the PRODUCT library's sampler against a test-build sampler with a switch set."""
    from mjhmc_amd import engine, _lib
    if device not in _HOOKS_CTX:
        _HOOKS_CTX[device] = engine.Context(device, lib=_lib.load_test_hooks())
    return _HOOKS_CTX[device]
