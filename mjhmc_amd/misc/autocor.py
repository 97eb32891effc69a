"""Sample-generation driver of the reference's analysis harness (mjhmc/misc/autocor.py:213-261), the
main caller of the hot path.  The reference performs one host round trip per step
(``samples[:, :, t] = smp.sample(1)`` and a read of the distribution's counters); here the whole run is
one batched device call and the per-step counter trace is rebuilt from the kernels' exact tallies."""
import numpy as np

from ..samplers.markov_jump_hmc import ContinuousTimeHMC


def generate_samples(sampler, distribution, num_steps=None, num_grad_steps=None, **kwargs):
    """Same contract as the reference:

    Returns (samples [n_dims, n_batch, n_samples], e_evals [n_samples], grad_evals [n_samples]) where
    ``grad_evals[t] = distribution.dEdX_count / n_batch`` after step t (counters reset after construction)."""
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
                return samples[:, :, :t_idx + 1], e_evals[:t_idx + 1], grad_evals[:t_idx + 1]
    else:
        samples = smp.sample(num_steps, preserve_order=True)              # one batched launch sequence
        trace = np.cumsum(smp.eval_trace(num_steps), axis=0)
        e_evals = trace[:, 0] / float(n_batch)
        grad_evals = trace[:, 1] / float(n_batch)
        if num_grad_steps is not None:
            hit = np.nonzero(grad_evals >= num_grad_steps)[0]
            if hit.size:                                                   # the reference stops at the first hit
                k = hit[0] + 1
                return samples[:, :, :k], e_evals[:k], grad_evals[:k]

    if num_grad_steps is not None:
        assert grad_evals[-1] >= num_grad_steps
        grad_sel = (grad_evals <= num_grad_steps)
        return samples[:, :, grad_sel], e_evals[grad_sel], grad_evals[grad_sel]
    return samples, e_evals, grad_evals
