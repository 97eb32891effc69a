#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- captures golden vectors from the *imported* reference.

Runs only in the build container (needs /root/reference); the GPU box never sees the reference.
The reference's NumPy hot path imports under Python 3 once ``xrange`` exists; its stock
``Distribution`` constructors are avoided because they pull in a py2-only module and start a
1e6-step burn-in (mjhmc/misc/distributions.py:104-149), so the harness subclasses
``Distribution`` with its own ``init_X`` and borrows ``E_val``/``dEdX_val`` unbound.

Every number the reference draws from ``np.random`` is recorded (``randn`` blocks, the *unit*
exponential behind each ``np.random.exponential(scale)`` call, ``rand``/``random`` uniforms)
together with the state after every ``sampling_iteration``.  Output: small ``.npz`` files under
tests/golden/.  Usage:  python oracle/capture_golden.py [--ref /root/reference]
"""
import argparse
import builtins
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')


def import_reference(path):
    builtins.xrange = range
    sys.path.insert(0, path)
    import mjhmc.samplers.markov_jump_hmc as ref_samplers
    import mjhmc.samplers.hmc_state as ref_state
    import mjhmc.misc.distributions as ref_distr
    import mjhmc.misc.utils as ref_utils
    return ref_samplers, ref_state, ref_distr, ref_utils


class Recorder(object):
    """Monkey-patches np.random and the reference's draw_from/min_idx bindings to log draws."""

    def __init__(self, ref_samplers, ref_utils, nbatch):
        self.rs, self.ru, self.n = ref_samplers, ref_utils, nbatch
        self.normals, self.exps, self.uniforms, self.which = [], [], [], []
        self._cur = None
        self._kind = 0

    def __enter__(self):
        self._o = (np.random.randn, np.random.exponential, np.random.rand, np.random.random,
                   self.rs.draw_from, self.rs.min_idx)
        o_randn, o_exp, o_rand, o_random, o_draw, o_min = self._o
        std_exp = np.random.standard_exponential
        rec = self

        def randn(*shape):
            z = o_randn(*shape)
            rec.normals.append(z.copy())
            return z

        def exponential(scale=1.0):
            e = std_exp()
            rec._pending.append(e)
            return scale * e

        def rand(*shape):
            u = o_rand(*shape)
            rec.uniforms.append(np.array(u, dtype=np.float64).copy())
            return u

        def random(*a):
            u = o_random(*a)
            rec.uniforms.append(np.array(u, dtype=np.float64).copy())
            return u

        def draw_from(rates):
            if rec._cur is None:
                rec._cur = np.full((3, rec.n), np.nan)
                rec._kind = 0
            rec._pending = []
            bad = ~np.isfinite(rates)
            stop = int(np.argmax(bad)) if bad.any() else len(rates)
            takers = [i for i in range(stop) if rates[i] != 0]
            try:
                out = o_draw(rates)
            finally:
                assert len(rec._pending) == len(takers)
                rec._cur[rec._kind, takers] = rec._pending
                rec._kind += 1
                if bad.any() or rec._kind == 3:
                    rec.exps.append(rec._cur)
                    rec._cur = None
            return out

        def min_idx(draws):
            out = o_min(draws)
            w = np.full(rec.n, 255, dtype=np.uint8)
            for k, idx in enumerate(out):
                w[idx] = k
            rec.which.append(w)
            return out

        np.random.randn, np.random.exponential = randn, exponential
        np.random.rand, np.random.random = rand, random
        self.rs.draw_from, self.rs.min_idx = draw_from, min_idx
        return self

    def __exit__(self, *exc):
        (np.random.randn, np.random.exponential, np.random.rand, np.random.random,
         self.rs.draw_from, self.rs.min_idx) = self._o
        return False


def make_harness(ref_distr, kind, X0, **par):
    Distribution = ref_distr.Distribution

    class Harness(Distribution):
        def __init__(self):
            for k, v in par.items():
                setattr(self, k, v)
            Distribution.__init__(self, X0.shape[0], X0.shape[1])

        def init_X(self):
            self.Xinit = X0

        def __hash__(self):
            return 0

    src = {'iso': ref_distr.TestGaussian, 'diag': ref_distr.Gaussian, 'rough': ref_distr.RoughWell,
           'mm': ref_distr.MultimodalGaussian}[kind]
    Harness.E_val = src.E_val
    Harness.dEdX_val = src.dEdX_val
    return Harness()


def energy_params(kind, ndims, nbatch):
    if kind == 'iso':
        return dict(sigma=1.3)
    if kind == 'diag':
        cond = 10 ** np.linspace(-2, 0, ndims)
        return dict(conditioning=cond, J=np.diag(cond))
    if kind == 'rough':
        return dict(scale1=100, scale2=4)
    if kind == 'mm':
        sep = 3
        sv = np.array([sep] * nbatch + [0] * (ndims - 1) * nbatch).reshape(ndims, nbatch)
        sv[0] += sep
        return dict(sep_vec=sv, separation=sep)
    raise KeyError(kind)


def snapshot(s, d):
    st = s.state
    return dict(X=st.X.copy(), V=st.V.copy(), EX=st.EX[0].copy(), EV=st.EV[0].copy(), dEdX=st.dEdX.copy(),
                cache=st.cache_active.copy() if hasattr(st, 'cache_active') else None,
                dwell=np.array(getattr(s, 'dwelling_times', np.zeros(s.nbatch))).copy(),
                counts=np.array([s.l_count, s.f_count, s.r_count, s.fl_count], dtype=np.int64),
                evals=np.array([d.E_count, d.dEdX_count], dtype=np.int64),
                hp=np.array([s.epsilon, s.num_leapfrog_steps], dtype=np.float64))


def stack(snaps, key):
    return np.stack([sn[key] for sn in snaps])


def capture_mjhmc(refs, name, kind, ndims, nbatch, eps, L, beta, T, seed, x_scale=1.0, keep_grad=True,
                  cls_name='MarkovJumpHMC'):
    rs, _, rd, ru = refs
    np.random.seed(seed)
    X0 = x_scale * np.random.randn(ndims, nbatch)
    par = energy_params(kind, ndims, nbatch)
    d = make_harness(rd, kind, X0, **par)
    with Recorder(rs, ru, nbatch) as rec:
        s = getattr(rs, cls_name)(distribution=d, epsilon=eps, beta=beta, num_leapfrog_steps=L, resample=False)
        snaps = [snapshot(s, d)]
        n_attempts = [len(rec.exps)]
        for _ in range(T):
            s.sampling_iteration()
            snaps.append(snapshot(s, d))
            n_attempts.append(len(rec.exps))
    out = dict(kind=kind, Xinit=X0, eps=eps, L=L, beta=beta, p_r=s.p_r, T=T,
               normals=np.stack(rec.normals), exps=np.stack(rec.exps), trans=np.stack(rec.which),
               attempts_done=np.array(n_attempts, dtype=np.int64))
    for k in ('X', 'V', 'EX', 'EV', 'dEdX', 'cache', 'dwell', 'counts', 'evals', 'hp'):
        if k != 'dEdX' or keep_grad:
            out[k] = stack(snaps, k)
    for k, v in par.items():
        if k not in ('J', 'sep_vec'):
            out['par_' + k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, 'attempts', len(rec.exps), 'counts', snaps[-1]['counts'], 'evals', snaps[-1]['evals'])


def capture_sample(refs, name, ndims, nbatch, eps, L, beta, n_samples, seed):
    rs, _, rd, ru = refs
    np.random.seed(seed)
    X0 = np.random.randn(ndims, nbatch)
    d = make_harness(rd, 'iso', X0, sigma=1.0)
    with Recorder(rs, ru, nbatch) as rec:
        s = rs.MarkovJumpHMC(distribution=d, epsilon=eps, beta=beta, num_leapfrog_steps=L)
        res = s.sample(n_samples)
    out = dict(kind='iso', par_sigma=np.asarray(1.0), Xinit=X0, eps=eps, L=L, beta=beta, n_samples=n_samples,
               normals=np.stack(rec.normals), exps=np.stack(rec.exps), trans=np.stack(rec.which),
               resample_u=rec.uniforms[-1], samples=res,
               counts=np.array([s.l_count, s.f_count, s.r_count, s.fl_count], dtype=np.int64),
               evals=np.array([d.E_count, d.dEdX_count], dtype=np.int64))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, res.shape, out['counts'])


def capture_control(refs, name, cls_name, kind, ndims, nbatch, eps, L, beta, T, seed):
    rs, _, rd, ru = refs
    np.random.seed(seed)
    X0 = np.random.randn(ndims, nbatch)
    par = energy_params(kind, ndims, nbatch)
    d = make_harness(rd, kind, X0, **par)
    with Recorder(rs, ru, nbatch) as rec:
        s = getattr(rs, cls_name)(distribution=d, epsilon=eps, beta=beta, num_leapfrog_steps=L)
        snaps = [snapshot(s, d)]
        n_norm = [len(rec.normals)]
        for _ in range(T):
            s.sampling_iteration()
            snaps.append(snapshot(s, d))
            n_norm.append(len(rec.normals))
    # uniforms per iteration: rand(N), rand(N), random()
    u_acc = np.stack(rec.uniforms[0::3])
    u_flip = np.stack(rec.uniforms[1::3])
    u_r = np.array([float(u) for u in rec.uniforms[2::3]])
    out = dict(kind=kind, cls=cls_name, Xinit=X0, eps=eps, L=L, beta_in=beta, beta=s.beta, p_r=s.p_r,
               p_flip=s.p_flip, T=T, normals=np.stack(rec.normals), normals_done=np.array(n_norm, dtype=np.int64),
               u_acc=u_acc, u_flip=u_flip, u_r=u_r)
    for k in ('X', 'V', 'EX', 'EV', 'dEdX', 'counts', 'evals'):
        out[k] = stack(snaps, k)
    for k, v in par.items():
        if k not in ('J', 'sep_vec'):
            out['par_' + k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    print(name, 'counts', snaps[-1]['counts'])


def capture_energies(refs):
    _, _, rd, _ = refs
    rng = np.random.RandomState(7)
    out = {}
    for kind, ndims, n in (('iso', 2, 100), ('iso', 512, 64), ('diag', 10, 33), ('rough', 5, 40), ('mm', 3, 20)):
        par = energy_params(kind, ndims, n)
        if kind == 'diag':
            cond = 10 ** np.linspace(-6, 0, ndims)
            par = dict(conditioning=cond, J=np.diag(cond))
        scale = 100.0 if kind == 'rough' else 1.5
        X = scale * rng.randn(ndims, n)
        d = make_harness(rd, kind, X, **par)
        tag = '%s_%dx%d' % (kind, ndims, n)
        out[tag + '_X'] = X
        out[tag + '_E'] = np.asarray(d.E_val(X)).reshape(-1)
        out[tag + '_g'] = d.dEdX_val(X)
        # gathered (F-ordered) operand, as the samplers pass it (hmc_state.py:47,53)
        idx = np.arange(0, n, 2)
        out[tag + '_Eg'] = np.asarray(d.E_val(X[:, idx])).reshape(-1)
        out[tag + '_gg'] = d.dEdX_val(X[:, idx])
    np.savez_compressed(os.path.join(OUT, 'g2_energies.npz'), **out)
    print('g2_energies', len(out))


def capture_min_idx(refs):
    ru = refs[3]
    np.random.seed(1)                           # mjhmc/tests/test_utils.py:5,13
    a, b = np.random.randn(100), np.random.randn(100)
    two = ru.min_idx([a.reshape(1, 100), b.reshape(1, 100)])
    np.random.seed(1)
    c, d, e = np.random.randn(100), np.random.randn(100), np.random.randn(100)
    three = ru.min_idx([c.reshape(1, 100), d.reshape(1, 100), e.reshape(1, 100)])
    # ties and infinities, as the jump process produces them (rate 0 -> inf wait)
    t = np.array([[1., np.inf, 2., np.inf, 0.5, 3.]])
    u = np.array([[1., np.inf, 1., 4., 0.5, np.inf]])
    v = np.array([[2., np.inf, 3., 4., 0.5, 1.]])
    ties = ru.min_idx([t, u, v])
    np.savez_compressed(os.path.join(OUT, 'g1_min_idx.npz'), a=a, b=b, two_0=two[0], two_1=two[1],
                        c=c, d=d, e=e, three_0=three[0], three_1=three[1], three_2=three[2],
                        t=t, u=u, v=v, ties_0=ties[0], ties_1=ties[1], ties_2=ties[2])
    print('g1_min_idx')


def capture_trajectories(refs):
    rs, rst, rd, _ = refs
    out = {}
    for tag, kind, ndims, n, eps, L in (('iso_2x100', 'iso', 2, 100, 0.1, 10), ('iso_512x32', 'iso', 512, 32, 0.05, 10),
                                        ('diag_16x24', 'diag', 16, 24, 0.4, 7), ('rough_4x16', 'rough', 4, 16, 0.5, 6)):
        np.random.seed(11)
        X0 = (100.0 if kind == 'rough' else 1.0) * np.random.randn(ndims, n)
        d = make_harness(rd, kind, X0, **energy_params(kind, ndims, n))
        s = rs.MarkovJumpHMC(distribution=d, epsilon=eps, beta=0.3, num_leapfrog_steps=L)
        V0 = s.state.V.copy()
        lz = s.state.copy().L()
        fz = s.state.copy().FLF()
        out[tag + '_X0'], out[tag + '_V0'] = X0, V0
        out[tag + '_hp'] = np.array([eps, L])
        out[tag + '_EX0'], out[tag + '_EV0'], out[tag + '_g0'] = s.state.EX[0].copy(), s.state.EV[0].copy(), s.state.dEdX.copy()
        for nm, z in (('L', lz), ('FLF', fz)):
            out['%s_%s_X' % (tag, nm)], out['%s_%s_V' % (tag, nm)] = z.X, z.V
            out['%s_%s_EX' % (tag, nm)], out['%s_%s_EV' % (tag, nm)] = z.EX[0], z.EV[0]
            out['%s_%s_g' % (tag, nm)] = z.dEdX
        if 'par_sigma' not in out and kind == 'iso':
            out['par_sigma'] = np.asarray(1.3)
    np.savez_compressed(os.path.join(OUT, 'g3_trajectories.npz'), **out)
    print('g3_trajectories')


def reference_autocor_functions(ref_root):
    """fft_autocor (:37-49), slow_autocorrelation (:177-211) and generate_samples (:213-261) of the reference's
    mjhmc/misc/autocor.py, executed from the reference's own file.  The module as a whole cannot be imported (Python 2
    print statements at :19-34, `from mklfft.fftpack import fftn, ifftn` at :5 with mklfft absent); these three
    functions are valid Python 3 on their own, and mklfft.fftpack.fftn / ifftn are numpy.fft's fftn / ifftn computed
    by MKL (same signature, same transform), so numpy.fft stands in for them."""
    lines = open(os.path.join(ref_root, 'mjhmc', 'misc', 'autocor.py')).read().split('\n')
    ns = {'np': np, 'fftn': np.fft.fftn, 'ifftn': np.fft.ifftn}
    for name in ('fft_autocor', 'slow_autocorrelation', 'generate_samples'):
        start = next(i for i, ln in enumerate(lines) if ln.startswith('def %s(' % name))
        stop = next((i for i in range(start + 1, len(lines)) if lines[i].startswith('def ')), len(lines))
        exec('\n'.join(lines[start:stop]), ns)
    return ns['fft_autocor'], ns['slow_autocorrelation'], ns['generate_samples']


def capture_autocor(refs, ref_root):
    """G9: the reference's main caller and the step right after it -- generate_samples (autocor.py:213-261) run on
    the imported samplers with every random number recorded, then fft_autocor / slow_autocorrelation of its output."""
    rs, _, rd, ru = refs
    fft_autocor, slow_autocorrelation, generate_samples = reference_autocor_functions(ref_root)
    for name, cls_name, kind, ndims, nbatch, eps, L, beta, T, seed in (
            ('g9_generate_mjhmc_diag_6x40', 'MarkovJumpHMC', 'diag', 6, 40, 0.5, 4, 0.4, 24, 601),
            ('g9_generate_control_iso_3x50', 'ControlHMC', 'iso', 3, 50, 0.3, 6, 0.6, 24, 602)):
        np.random.seed(seed)
        X0 = np.random.randn(ndims, nbatch)
        par = energy_params(kind, ndims, nbatch)
        d = make_harness(rd, kind, X0, **par)
        kw = dict(epsilon=eps, beta=beta, num_leapfrog_steps=L)
        if cls_name == 'MarkovJumpHMC':
            kw['resample'] = False
        with Recorder(rs, ru, nbatch) as rec:
            samples, e_evals, grad_evals = generate_samples(getattr(rs, cls_name), d, num_steps=T, **kw)
        out = dict(kind=kind, cls=cls_name, Xinit=X0, eps=eps, L=L, beta_in=beta, T=T, normals=np.stack(rec.normals),
                   samples=samples, e_evals=e_evals, grad_evals=grad_evals, fft_autocor=fft_autocor(samples),
                   slow_autocor=slow_autocorrelation(samples, e_evals, grad_evals, half_window=False)[0])
        if cls_name == 'MarkovJumpHMC':
            assert len(rec.exps) == T, 'a retry happened: pick a milder step size'
            out['exps'] = np.stack(rec.exps)
        else:
            out['u_acc'] = np.stack(rec.uniforms[0::3])
            out['u_flip'] = np.stack(rec.uniforms[1::3])
            out['u_r'] = np.array([float(u) for u in rec.uniforms[2::3]])
            smp = getattr(rs, cls_name)(distribution=make_harness(rd, kind, X0, **par), **kw)
            out['p_r'], out['beta'], out['p_flip'] = smp.p_r, smp.beta, smp.p_flip
        for k, v in par.items():
            if k not in ('J', 'sep_vec'):
                out['par_' + k] = np.asarray(v)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
        print(name, samples.shape, 'grad_evals[-1]', grad_evals[-1], 'autocor[1]', out['fft_autocor'][1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--only', default='', help='comma separated fixture names to (re)generate')
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    refs = import_reference(args.ref)
    np.seterr(all='ignore')
    only = set(filter(None, args.only.split(',')))
    if only:
        real_savez = np.savez_compressed

        def filtered(path, **kw):
            if os.path.splitext(os.path.basename(path))[0] in only:
                real_savez(path, **kw)
        np.savez_compressed = filtered
    capture_min_idx(refs)
    capture_energies(refs)
    capture_trajectories(refs)
    # G4: full sampling_iteration replays, three hyper-parameter sets + other energies
    capture_mjhmc(refs, 'g4_iso_2x100_a', 'iso', 2, 100, 0.1, 10, 0.1, 20, 101)
    capture_mjhmc(refs, 'g4_iso_2x100_b', 'iso', 2, 100, 1.0, 10, 0.8, 20, 102)
    capture_mjhmc(refs, 'g4_diag_16x64', 'diag', 16, 64, 0.9, 5, 0.3, 20, 103)
    capture_mjhmc(refs, 'g4_iso_512x32', 'iso', 512, 32, 0.05, 10, 0.1, 5, 104, keep_grad=False)
    capture_mjhmc(refs, 'g4_rough_4x48', 'rough', 4, 48, 0.5, 8, 0.2, 20, 105, x_scale=100.0)
    capture_mjhmc(refs, 'g4_mm_3x40', 'mm', 3, 40, 0.3, 6, 0.4, 20, 106)
    capture_mjhmc(refs, 'g4_iso_33x17', 'iso', 33, 17, 0.3, 3, 0.5, 12, 107)
    # G5: README-shaped sample(10) with dwell-time resampling
    capture_sample(refs, 'g5_sample_2x100', 2, 100, 0.3, 5, 0.3, 10, 201)
    # G6: non-finite rate -> halve eps / double L / retry (markov_jump_hmc.py:376-389)
    capture_mjhmc(refs, 'g6_retry_a_iso_4x32', 'iso', 4, 32, 1.0, 4, 0.3, 6, 301, x_scale=900.0)
    capture_mjhmc(refs, 'g6_retry_b_iso_4x32', 'iso', 4, 32, 1.5, 4, 0.3, 6, 301, x_scale=500.0)
    # G7: discrete-time control samplers
    capture_control(refs, 'g7_control_iso_2x100', 'ControlHMC', 'iso', 2, 100, 0.4, 5, 0.5, 12, 401)
    capture_control(refs, 'g7_hmc_diag_8x32', 'HMC', 'diag', 8, 32, 0.5, 4, 0.4, 12, 402)
    capture_control(refs, 'g7_base_iso_3x50', 'HMCBase', 'iso', 3, 50, 0.3, 6, 0.6, 12, 403)
    # G8: ContinuousTimeHMC (F / FL / R clocks); trans rows are min_idx's [f, fl, r] order
    capture_mjhmc(refs, 'g8_cthmc_diag_6x40', 'diag', 6, 40, 0.5, 4, 0.4, 15, 501, cls_name='ContinuousTimeHMC')
    # G9: generate_samples + autocorrelation, from the reference's own (extracted) functions
    capture_autocor(refs, args.ref)


if __name__ == '__main__':
    main()
