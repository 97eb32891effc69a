"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's autocorrelation estimators.

Follows mjhmc/misc/autocor.py: fft_autocor (:37-49), the lag-product estimator of the brute-force branch
of autocorrelation (:52-117, with compile_autocor_func :119-140) and slow_autocorrelation (:177-211).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Parity status: pinned through fixtures G9 (tests/golden/g9_generate_*.npz).  mjhmc/misc/autocor.py cannot be
imported as a module (Python 2 print statements at :19, mklfft absent), and the one reference test of this code
(tests/test_fast_ac.py) drives a DataFrame API that no longer exists in it; oracle/capture_golden.py therefore
executes the reference's fft_autocor, slow_autocorrelation and generate_samples FROM THE REFERENCE'S FILE (each is
valid Python 3 on its own; numpy.fft stands in for mklfft's identical fftn / ifftn) on a recorded run of the
imported samplers.  test_oracle_golden.py checks this restatement against those outputs, and against the explicit
lag-sum definitions."""
import numpy as np


def fft_autocor(samples):
    """autocor.py:37-49 with numpy.fft in place of mklfft (same transform, same axis)."""
    assert samples.ndim == 3
    fft_samples = np.fft.fftn(samples, axes=[-1])
    fft_ac = np.real(np.mean(np.fft.ifftn(fft_samples * np.conj(fft_samples), axes=[-1]), axis=(0, 1)))
    return fft_ac / fft_ac[0]


def circular_lag_sums(samples):
    """Definition behind fft_autocor: out[k] = sum_{d,n} sum_t x[t] x[(t+k) mod T] (O(T^2), small inputs)."""
    T = samples.shape[2]
    return np.array([np.sum(samples * np.roll(samples, -k, axis=2)) for k in range(T)])


def linear_lag_sums(samples):
    """out[k] = sum_{d,n} sum_{t < T-k} x[t] x[t+k]."""
    T = samples.shape[2]
    return np.array([np.sum(samples[:, :, :T - k] * samples[:, :, k:]) for k in range(T)])


def slow_autocorrelation(samples, e_evals, grad_evals, half_window=False):
    """autocor.py:177-211 (``T/2`` is integer division in the reference's Python 2)."""
    _, _, T = samples.shape
    n_lags = T - 1 if not half_window else (T // 2) - 1
    c = np.zeros((n_lags,))
    c[0] = np.mean(samples ** 2)
    for t_gap in range(1, n_lags):
        c[t_gap] = np.mean(samples[:, :, :-t_gap] * samples[:, :, t_gap:])
    return c / c[0], e_evals, grad_evals


def autocorrelation(samples, e_evals, grad_evals, half_window=True, normalize=True, cached_var=None,
                    brute_force=False):
    """autocor.py:52-117; the brute-force branch restates the theano scan of :119-140."""
    _, _, n_samples = samples.shape
    if brute_force:
        max_t = (n_samples // 2) - 1 if half_window else n_samples - 1
        ac_squeeze = np.array([np.mean(samples[:, :, :-t] * samples[:, :, t:]) for t in range(1, max_t)])
        var = np.mean(samples ** 2, keepdims=True)[0][0] if cached_var is None else cached_var
        if normalize:
            autocor = np.vstack((1., (ac_squeeze / var).reshape(-1, 1)))
        else:
            autocor = np.vstack((var, ac_squeeze.reshape(-1, 1)))
        if half_window:
            e_evals = e_evals[:int(n_samples / 2) - 1]
            grad_evals = grad_evals[:int(n_samples / 2) - 1]
        else:
            e_evals = e_evals[:-1]
            grad_evals = grad_evals[:-1]
    else:
        autocor = fft_autocor(samples)
        assert autocor.shape == e_evals.shape
        assert e_evals.shape == grad_evals.shape
    return autocor, e_evals, grad_evals
