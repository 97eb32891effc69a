"""Analysis harness around the hot path (mjhmc/misc/autocor.py): the sample-generation driver
(:213-261) and the autocorrelation that every experiment computes from its output (:37-117, :177-211).

The reference performs one host round trip per step (``samples[:, :, t] = smp.sample(1)`` and a read of the
distribution's counters) and then transforms the (n_dims, n_batch, n_samples) host array with mklfft.
Here the whole run is one batched device call, the per-step counter trace is rebuilt from the kernels'
exact tallies, and the autocorrelation is taken from the device-resident sample ring (batched hipFFT
between two fused kernels, mjhmc_amd/csrc/autocor.hip) - the samples only travel to the host if the
caller asks for them."""
import numpy as np

from .. import engine
from ..samplers.markov_jump_hmc import ContinuousTimeHMC


def fft_autocor(samples, device=0):
    """mjhmc/misc/autocor.py:37-49: autocorrelation by the cross-correlation theorem.

    samples: [n_dims, n_batch, n_samples] -> autocor [n_samples], normalised to 1 at lag 0."""
    samples = np.asarray(samples)
    assert samples.ndim == 3
    sums = engine.context(device).autocor(samples, linear=False)
    return sums / sums[0]


def _lag_means(sums, n_series, n_samples):
    """np.mean(samples[:, :, :-k] * samples[:, :, k:]) for every lag k from the linear lag sums."""
    return sums / (float(n_series) * (n_samples - np.arange(n_samples)))


def slow_autocorrelation(samples, e_evals, grad_evals, half_window=False, device=0):
    """mjhmc/misc/autocor.py:177-211 (zero-mean assumption, window truncated like the reference)."""
    n_dims, n_batch, n_samples = samples.shape
    c = _lag_means(engine.context(device).autocor(samples, linear=True), n_dims * n_batch, n_samples)
    n_lags = (n_samples // 2) - 1 if half_window else n_samples - 1
    c = c[:n_lags]
    return c / c[0], e_evals, grad_evals


def autocorrelation(samples, e_evals, grad_evals, half_window=True, normalize=True, cached_var=None,
                    brute_force=False, use_tf=False, device=0):
    """mjhmc/misc/autocor.py:52-117.  ``brute_force`` selects the lag-product estimator the reference
    compiles with theano / tensorflow (``use_tf`` only chose between those two and is ignored)."""
    n_dims, n_batch, n_samples = samples.shape
    if brute_force:
        c = _lag_means(engine.context(device).autocor(samples, linear=True), n_dims * n_batch, n_samples)
        max_t = (n_samples // 2) - 1 if half_window else n_samples - 1
        ac_squeeze = c[1:max_t]
        var = c[0] if cached_var is None else cached_var
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
        autocor = fft_autocor(samples, device=device)
        assert autocor.shape == e_evals.shape
        assert e_evals.shape == grad_evals.shape
    return autocor, e_evals, grad_evals


def calculate_autocorrelation(sampler, distribution, num_steps=None, num_grad_steps=None, sample_steps=1,
                              half_window=False, use_cached_var=False, replay=None, **kwargs):
    """mjhmc/misc/autocor.py:10-35.  Returns (autocor, e_evals, grad_evals).

    When the run is recorded in the device ring (every sampler except a resampling jump sampler) the
    autocorrelation is computed there and the (n_dims, n_batch, n_samples) block is never downloaded."""
    smp, samples, e_evals, grad_evals, n_keep = _generate(sampler, distribution, num_steps, num_grad_steps,
                                                          download=False, replay=replay, **kwargs)
    if samples is not None:
        return autocorrelation(samples, e_evals, grad_evals, half_window)
    sums = smp._dev.ring_autocor(0, n_keep, linear=False)
    if smp._comm is not None:
        sums = smp._comm.allreduce_f64(sums)          # column shards add their lag sums
    autocor = sums / sums[0]
    assert autocor.shape == e_evals.shape
    return autocor, e_evals, grad_evals


def generate_samples(sampler, distribution, num_steps=None, num_grad_steps=None, replay=None, **kwargs):
    """Same contract as the reference:

    Returns (samples [n_dims, n_batch, n_samples], e_evals [n_samples], grad_evals [n_samples]) where
    ``grad_evals[t] = distribution.dEdX_count / n_batch`` after step t (counters reset after construction).
    ``replay``: recorded random numbers, one entry per iteration as ``sampling_iteration(replay=...)`` takes them
    (parity tests against the reference's own generate_samples, tests/golden/g9_*)."""
    _, samples, e_evals, grad_evals, _ = _generate(sampler, distribution, num_steps, num_grad_steps,
                                                   download=True, replay=replay, **kwargs)
    return samples, e_evals, grad_evals


def _generate(sampler, distribution, num_steps, num_grad_steps, download, replay=None, **kwargs):
    """(sampler, samples or None, e_evals, grad_evals, n_keep).  ``samples`` is None when ``download`` is
    false and ring slots [0, n_keep) hold the run."""
    assert (((num_steps is None) and (num_grad_steps is not None)) or
            (num_steps is not None) and (num_grad_steps is None))
    smp = sampler(distribution=distribution, **kwargs)
    num_steps = int(num_steps or num_grad_steps / smp.grad_per_sample_step + 100)
    n_batch = distribution.nbatch
    # the reference calls distribution.reset() here: counters to zero (its init_X re-run does not touch
    # the already built sampler state)
    distribution.E_count = 0
    distribution.dEdX_count = 0

    if isinstance(smp, ContinuousTimeHMC) and smp.resample:
        # sample(1) with dwell-time resampling is two iterations plus a host-side draw per step: keep the
        # reference's step structure literally (each step is still device work)
        n_dims = distribution.ndims
        samples = np.zeros((n_dims, n_batch, num_steps))
        grad_evals = np.zeros(num_steps)
        e_evals = np.zeros(num_steps)
        for t_idx in range(num_steps):
            samples[:, :, t_idx] = smp.sample(1)
            grad_evals[t_idx] = distribution.dEdX_count / float(n_batch)
            e_evals[t_idx] = distribution.E_count / float(n_batch)
            if (num_grad_steps is not None) and grad_evals[t_idx] >= num_grad_steps:
                k = t_idx + 1
                return smp, samples[:, :, :k], e_evals[:k], grad_evals[:k], k
        if num_grad_steps is not None:
            assert grad_evals[-1] >= num_grad_steps
        return smp, samples, e_evals, grad_evals, num_steps

    smp._record(num_steps, replay)                                         # one batched launch sequence
    trace = np.cumsum(smp.eval_trace(num_steps), axis=0)
    e_evals = trace[:, 0] / float(n_batch)
    grad_evals = trace[:, 1] / float(n_batch)
    n_keep = num_steps
    if num_grad_steps is not None:
        hit = np.nonzero(grad_evals >= num_grad_steps)[0]
        assert hit.size, 'the run ended before num_grad_steps gradient evaluations per particle'
        n_keep = int(hit[0]) + 1                                           # the reference stops at the first hit
    samples = smp._stack(n_keep, True) if download else None
    return smp, samples, e_evals[:n_keep], grad_evals[:n_keep], n_keep
