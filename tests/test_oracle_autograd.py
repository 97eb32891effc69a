"""CPU: the oracle's HAND-DERIVED gradients of the energies the reference builds symbolically (ProductOfT via Theano,
Funnel / SparseImageCode via TensorFlow -- none importable) against an independent torch restatement of the reference's
FORWARD graphs with autograd gradients (mirrors T.grad, distributions.py:408, and tf.gradients,
tf_distributions.py:91), live and through the committed G2 fixtures (oracle/capture_dense_fixtures.py)."""
import numpy as np
import pytest

from oracle import mjhmc_oracle as orc
from oracle import autograd_energies as ag
from tests.helpers import load, ref_init_weights, sic_problem

TOL = 1e-12


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize('D,n', [(36, 25), (100, 7), (512, 16)])
def test_product_of_t_gradient_is_autograd(D, n):
    W, lognu = ref_init_weights(D, D)
    b = 0.1 * np.random.RandomState(D).randn(D)
    X = np.random.RandomState(n).randn(D, n) * 1.5
    o = orc.ProductOfT(W, lognu=lognu, b=b, force_dtype=np.float64)
    E, g = ag.product_of_t(W, np.exp(lognu), b, X)
    assert rel(o.E_val(X)[0], E) < TOL and rel(o.dEdX_val(X), g) < TOL


@pytest.mark.parametrize('scale', [1.0, 3.0])
def test_funnel_gradients_are_autograd(scale):
    X = np.random.RandomState(3).randn(10, 50)
    X[0] *= 2.0
    E, g = ag.funnel_literal(scale, X)
    o = orc.FunnelLiteral(scale)
    assert rel(o.E_val(X), E) < TOL and rel(o.dEdX_val(X), g) < TOL
    E, g = ag.funnel_neal(scale, X)
    o = orc.FunnelNeal(scale)
    assert rel(o.E_val(X)[0], E) < TOL and rel(o.dEdX_val(X), g) < TOL


@pytest.mark.parametrize('P,cauchy', [(1, True), (1, False), (3, True), (9, True)])
def test_sparse_image_code_gradient_is_autograd(P, cauchy):
    B, imgs, a0 = sic_problem(1, n_patches=P, img=64, n_coeffs=128)
    patches = imgs[:, :P].T
    o = orc.SparseImageCode(B, patches, lmbda=0.01, cauchy=cauchy)
    # one active column: the reference graph exactly as coded (tf_distributions.py:241-272)
    x1 = a0[:, None] + 0.3 * np.random.RandomState(2).randn(P * 128, 1)
    E, g = ag.sparse_image_code_literal(B, patches, 0.01, cauchy, x1)
    assert rel(o.E_val(x1)[0], E) < TOL and rel(o.dEdX_val(x1), g) < TOL
    # several columns: the graph column by column == the per-particle maths of the oracle
    X = a0[:, None] + 0.3 * np.random.RandomState(4).randn(P * 128, 5)
    E, g = ag.sparse_image_code_per_column(B, patches, 0.01, cauchy, X)
    assert rel(o.E_val(X)[0], E) < TOL and rel(o.dEdX_val(X), g) < TOL


def test_sparse_image_code_reshape_interleaves_beyond_one_column():
    """for more than one active column the reference's reshape mixes particles and coefficients, so
    the literal graph differs from the per-particle maths (which is what the engine implements)."""
    B, imgs, a0 = sic_problem(1, n_patches=2, img=64, n_coeffs=128)
    X = a0[:, None] + 0.3 * np.random.RandomState(4).randn(256, 3)
    E_lit, _ = ag.sparse_image_code_literal(B, imgs[:, :2].T, 0.01, True, X)
    E_col, _ = ag.sparse_image_code_per_column(B, imgs[:, :2].T, 0.01, True, X)
    assert rel(E_lit, E_col) > 1e-3


# ---- the committed fixtures -------------------------------------------------------------------------------------
def dense_cases():
    g = load('g2_dense')
    for D, n in ((36, 25), (512, 64)):
        tag = 'pot_%dx%d' % (D, n)
        W, lognu = ref_init_weights(D, D)
        yield tag, g, orc.ProductOfT(W, lognu=lognu, b=g[tag + '_b'], force_dtype=np.float64)
    for s in (1, 3):
        yield 'funnel_lit_s%d_10x50' % s, g, orc.FunnelLiteral(float(s))
    for tag in ('funnel_neal_s3_10x50', 'funnel_neal_s3_32x64'):
        yield tag, g, orc.FunnelNeal(3.0)
    for P, n, cauchy in ((1, 1, True), (1, 8, True), (1, 8, False), (9, 1, True), (9, 4, True)):
        B, imgs, _ = sic_problem(0, n_patches=P)
        yield ('sic_p%d_n%d_%s' % (P, n, 'cauchy' if cauchy else 'laplace'), g,
               orc.SparseImageCode(B, imgs[:, :P].T, lmbda=0.01, cauchy=cauchy))
    r = load('ref_init_states')
    W, lognu = ref_init_weights(36, 36)
    yield 'pot36', r, orc.ProductOfT(W, lognu=lognu, force_dtype=np.float64)
    B, imgs, _ = sic_problem(0)
    for tag in ('sic_mj', 'sic_ctl'):
        yield tag, r, orc.SparseImageCode(B, imgs[:, :1].T, lmbda=0.01, cauchy=True)


@pytest.mark.parametrize('tag,g,o', list(dense_cases()), ids=[c[0] for c in dense_cases()])
def test_oracle_matches_dense_fixtures(tag, g, o):
    X = g[tag + '_X']
    assert rel(np.asarray(o.E_val(X)).reshape(-1), g[tag + '_E']) < TOL
    assert rel(o.dEdX_val(X), g[tag + '_g']) < TOL


def test_recipes_reproduce_the_fixture_models():
    """The fixtures hold only inputs and outputs of the big models; the matrices come from seeded recipes."""
    import hashlib
    g = load('g2_dense')

    def digest(a):
        return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a, dtype=np.float64).tobytes()).digest()[:8], dtype=np.uint64)[0]
    assert digest(ref_init_weights(512, 512)[0]) == g['pot_512x64_Wdigest']
    assert np.array_equal(ref_init_weights(36, 36)[0], g['pot_36x25_W'])
    assert digest(sic_problem(0)[0]) == g['sic_p1_n8_cauchy_Bdigest']
    assert digest(sic_problem(0, n_patches=9)[0]) == g['sic_p9_n4_cauchy_Bdigest']


def test_float32_graph_is_within_float32_of_the_float64_one():
    """The reference evaluates ProductOfT in float32 (allow_input_downcast, distributions.py:413-415); the float32
    fixture values bound what 'float32 tolerance' means for the device kernel's parity bars."""
    g = load('g2_dense')
    for tag in ('pot_36x25', 'pot_512x64'):
        assert rel(g[tag + '_E32'], g[tag + '_E']) < 5e-6
        assert rel(g[tag + '_g32'], g[tag + '_g']) < 2e-5


def test_reference_written_initialisation_files_load():
    """The cache files the reference ships (Python 2 pickles, initializations/*.pickle) through the product's loader;
    their content must be what oracle/capture_dense_fixtures.py committed as ref_init_states.npz.  Runs where the
    reference checkout exists (the build container); the GPU box has only the .npz."""
    import os
    from mjhmc_amd.misc.gen_mj_init import load_reference_initialization
    root = '/root/reference/initializations'
    if not os.path.isdir(root):
        pytest.skip('no reference checkout here')
    r = load('ref_init_states')
    mj, emc_var, true_var, ctl = load_reference_initialization(os.path.join(root, 'ProductOfT_6123388416598428958.pickle'))
    assert mj.shape == (36, 1000) and ctl is None and emc_var > 0 and true_var > 0
    assert np.array_equal(mj[:, :256], r['pot36_X'])
    mj, emc_var, true_var, ctl = load_reference_initialization(
        os.path.join(root, 'SparseImageCode_-2828851975638192263.pickle'))
    assert np.array_equal(mj, r['sic_mj_X']) and np.array_equal(ctl, r['sic_ctl_X'])
    mj, _, _, _ = load_reference_initialization(os.path.join(root, 'Funnel_3713081631925750456.pickle'))
    assert mj.shape == (10, 1000) and np.isnan(mj).any()          # the shipped funnel run diverged


def test_initialisation_cache_loader_resolves_arrays_only(tmp_path):
    """The cache files are pickles; the loaders resolve ndarray / dtype reconstruction and nothing else, so a file that
    names another callable is refused instead of executed."""
    import os
    import pickle
    import numpy as np
    from mjhmc_amd.misc.gen_mj_init import load_reference_initialization

    class Evil(object):
        def __reduce__(self):
            return (os.system, ('echo should-never-run > %s' % (tmp_path / 'ran'),))
    bad = tmp_path / 'bad.pickle'
    bad.write_bytes(pickle.dumps((Evil(), 1.0, 2.0), protocol=2))
    with pytest.raises(pickle.UnpicklingError):
        load_reference_initialization(str(bad))
    assert not (tmp_path / 'ran').exists()
    good = tmp_path / 'good.pickle'
    good.write_bytes(pickle.dumps((np.arange(6.0).reshape(2, 3), np.float64(1.5), 2.5, np.ones((2, 3))), protocol=2))
    mj, a, b, ctl = load_reference_initialization(str(good))
    assert mj.shape == (2, 3) and (a, b) == (1.5, 2.5) and ctl.shape == (2, 3)


def test_initialisation_cache_round_trip_over_pickle_protocols_and_opaque_callables(tmp_path):
    """The library reads back what it (or any pickle protocol's array encoding) writes, and the stable cache key exists
    for a LambdaDistribution of opaque callables too (named by its `name`, as in the reference's __hash__)."""
    import pickle
    from mjhmc_amd.misc import gen_mj_init as G
    payload = (np.arange(6.0).reshape(2, 3), np.float64(1.5), 2.5, np.ones((2, 3)))
    for proto in range(2, pickle.HIGHEST_PROTOCOL + 1):
        f = tmp_path / ('p%d.pickle' % proto)
        f.write_bytes(pickle.dumps(payload, protocol=proto))
        mj, a, b, ctl = G.load_reference_initialization(str(f))
        assert np.array_equal(mj, payload[0]) and (a, b) == (1.5, 2.5) and np.array_equal(ctl, payload[3]), proto

    class Opaque(object):                       # what stable_digest reads of a distribution
        ndims, name = 7, 'my-energy'

        def device_energy(self):
            from mjhmc_amd import _lib
            return _lib.E_HOST, (lambda X: X.sum(axis=0), lambda X: np.ones_like(X))
    d1, d2 = Opaque(), Opaque()
    d2.name = 'another'
    assert len(G.stable_digest(d1)) == 16 and G.stable_digest(d1) == G.stable_digest(Opaque()) != G.stable_digest(d2)
