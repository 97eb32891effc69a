"""mjhmc.misc.utils on the device (mjhmc_min_idx / mjhmc_draw_from): the reference's own tests of min_idx
(mjhmc/tests/test_utils.py:15-52) against the committed fixture and NumPy controls, and draw_from against the
reference's formula with the same np.random stream (mjhmc/misc/utils.py:31-50)."""
import numpy as np
import pytest

from tests.helpers import load

pytestmark = pytest.mark.gpu


def test_min_idx_two_and_three_case_like_the_reference():
    from mjhmc_amd.misc.utils import min_idx
    np.random.seed(1)
    n = 100
    l1, l2, l3 = np.random.randn(n), np.random.randn(n), np.random.randn(n)
    m1, m2 = min_idx([l1.reshape(1, n), l2.reshape(1, n)])
    assert np.array_equal(m1, np.arange(n)[l1 < l2]) and np.array_equal(m2, np.arange(n)[l1 >= l2])
    a, b, c = min_idx([l1.reshape(1, n), l2.reshape(1, n), l3.reshape(1, n)])
    which = np.argmin(np.stack([l1, l2, l3]), axis=0)
    for i, got in enumerate((a, b, c)):
        assert np.array_equal(got, np.where(which == i)[0])
    assert set(a) == set(np.arange(n)[l1 < l2]) & set(np.arange(n)[l1 < l3])


def test_min_idx_fixture_ties_and_nan():
    from mjhmc_amd.misc.utils import min_idx
    g = load('g1_min_idx')                                    # the reference's own min_idx on these inputs (oracle/capture_golden.py)
    i0, i1 = min_idx([g['a'].reshape(1, -1), g['b'].reshape(1, -1)])
    assert np.array_equal(i0, g['two_0']) and np.array_equal(i1, g['two_1'])
    j = min_idx([g[k].reshape(1, -1) for k in 'cde'])
    assert all(np.array_equal(j[i], g['three_%d' % i]) for i in range(3))
    t = min_idx([g['t'], g['u'], g['v']])
    assert all(np.array_equal(t[i], g['ties_%d' % i]) for i in range(3))
    d = np.array([[1.0, 2.0, np.nan, 0.5, 3.0, np.inf],
                  [1.0, 1.0, 0.0, np.nan, 3.0, np.inf],
                  [0.0, 1.0, 0.0, 0.0, 3.0, np.inf]])
    got = min_idx([d[0:1], d[1:2], d[2:3]])
    which = np.argmin(d, axis=0)                              # first minimum; the first NaN wins
    for i in range(3):
        assert np.array_equal(got[i], np.where(which == i)[0])
    big = np.random.RandomState(3).randn(4, 100003)
    got = min_idx([big[i:i + 1] for i in range(4)])
    which = np.argmin(big, axis=0)
    for i in range(4):
        assert np.array_equal(got[i], np.where(which == i)[0])


def test_draw_from_is_numpys_exponential_bit_for_bit():
    from mjhmc_amd.misc.utils import draw_from
    rs = np.random.RandomState(5)
    rates = np.exp(rs.randn(1000) * 3)
    rates[[3, 500, 999]] = 0.0
    np.random.seed(11)
    want = np.array([np.inf if r == 0 else np.random.exponential(scale=1. / r) for r in rates]).reshape(1, -1)
    after_ref = np.random.random()
    np.random.seed(11)
    got = draw_from(rates)
    assert got.shape == (1, 1000) and np.array_equal(got, want)
    assert np.random.random() == after_ref                    # the same number of draws was consumed
    assert draw_from(np.zeros(0)).shape == (1, 0)


def test_draw_from_raises_like_the_reference_and_leaves_the_stream_where_it_does():
    from mjhmc_amd.misc.utils import draw_from
    rates = np.array([1.0, 0.0, 2.0, np.inf, 3.0, np.nan])
    np.random.seed(2)
    for r in rates[:3]:
        if r != 0:
            np.random.exponential(scale=1. / r)
    after_ref = np.random.random()
    np.random.seed(2)
    with pytest.raises(ValueError, match='Infinite rate'):
        draw_from(rates)
    assert np.random.random() == after_ref


def test_draw_from_negative_rate_raises_numpys_error_at_the_reference_position():
    """np.random.exponential(scale=1 / rate) with rate < 0 raises ValueError('scale < 0') inside the reference's loop
    (utils.py:31-49), after the draws of the rates in front of it."""
    from mjhmc_amd.misc.utils import draw_from
    rates = np.array([0.5, 0.0, 2.0, -1.0, 3.0, np.inf])
    np.random.seed(9)
    with pytest.raises(ValueError, match='scale < 0'):
        draw_from(rates)
    after = np.random.standard_exponential()
    np.random.seed(9)
    with pytest.raises(ValueError, match='scale < 0'):
        for r in rates:                       # the reference's loop
            if r == 0:
                continue
            if not np.isfinite(r):
                raise ValueError('Infinite rate')
            np.random.exponential(scale=1. / r)
    assert after == np.random.standard_exponential()
