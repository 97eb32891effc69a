"""mjhmc.misc.utils for callers of the hot path's two helpers (mjhmc/misc/utils.py): `min_idx` and `draw_from` run on the
device through the C ABI (mjhmc_min_idx / mjhmc_draw_from: the very device functions the samplers' kernels use for
the first minimum and the waiting times); the samplers themselves never come through here -- their jump process is
fused into one kernel.  `normalize_by_row` belongs to the algebraic ladder samplers (out of scope)."""
import numpy as np

from .. import engine


def overrides(interface_class):
    """Decorator asserting that the method exists in `interface_class` (utils.py:5-12)."""
    def overrider(method):
        assert method.__name__ in dir(interface_class)
        return method
    return overrider


def min_idx(draws, device=0):
    """For each of the draws (arrays of shape (1, N)) the indices of the columns where it is the minimum: the argmin
    down the stacked rows, first minimum on ties (utils.py:15-28)."""
    cdraws = np.concatenate([np.asarray(d, dtype=np.float64) for d in draws], axis=0)
    which = engine.context(device).min_idx(cdraws)
    return [np.where(which == i)[0] for i in range(len(draws))]


def draw_from(rates, device=0):
    """Draws from exponential distributions with the given rates, shape (1, len(rates)) (utils.py:31-50): inf where a
    rate is zero; a non-finite rate raises ValueError, and so does a negative one (np.random.exponential's own
    "scale < 0", which is what the reference's loop runs into).  np.random's stream is consumed exactly as the
    reference consumes it -- one standard exponential per finite non-zero rate, in order, up to the first rate that
    raises -- and the draws are bit-identical to np.random.exponential(scale=1. / rate)."""
    rates = np.asarray(rates, dtype=np.float64)
    assert rates.ndim == 1
    finite = np.isfinite(rates)
    raises = ~finite | (rates < 0)
    stop = int(np.argmax(raises)) if raises.any() else rates.size           # the reference raises there
    takes = finite & (rates != 0)
    takes[stop:] = False
    e = np.zeros(rates.size)
    e[takes] = np.random.standard_exponential(int(takes.sum()))
    if stop < rates.size and finite[stop]:
        raise ValueError('scale < 0')                                       # numpy's message for exponential(scale=1 / rate), rate < 0
    out, first_bad = engine.context(device).draw_from(rates, e)
    if first_bad >= 0:
        raise ValueError("Infinite rate. This occurs when calculating transition rates "
                         "between states that have a very large energy difference, such that "
                         "the transition probability is less than the numerical precision. "
                         "Try decreasing the leapfrog stepsize/number of steps or dividing "
                         " the energy by a large constant.")
    return out.reshape(1, rates.size)


def package_path():
    """Absolute path of the directory that holds this package (utils.py:60-70 looks 'MJHMC' up in sys.path)."""
    import os
    return os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
