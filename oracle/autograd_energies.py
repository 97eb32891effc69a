"""TEST INFRASTRUCTURE ONLY -- torch (CPU, float64) restatements of the reference's *symbolic* energies.

ProductOfT, Funnel and SparseImageCode are Theano / TensorFlow-0.x graphs in the reference; neither package exists
here, so their forward expressions are restated operation for operation in torch and the gradient is left to
autograd, exactly as the reference leaves it to ``T.grad`` (mjhmc/misc/distributions.py:408) and ``tf.gradients``
(mjhmc/misc/tf_distributions.py:91).  This is an INDEPENDENT check of the hand-derived gradients in
oracle/mjhmc_oracle.py (tests/test_oracle_autograd.py) and the generator of the G2 dense fixtures
(oracle/capture_dense_fixtures.py).  The product never imports it.
"""
import numpy as np
import torch


def _t(a, dtype=torch.float64):
    return torch.as_tensor(np.asarray(a, dtype=np.float64)).to(dtype)


def _with_grad(fn, X, dtype=torch.float64):
    x = _t(X, dtype).clone().requires_grad_(True)
    e = fn(x)
    (g,) = torch.autograd.grad(e.sum(), x)
    return e.detach().double().numpy().reshape(-1), g.detach().double().numpy()


def product_of_t(W, nu, b, X, dtype=torch.float64):
    """E_def of mjhmc/misc/distributions.py:420-433 and T.grad(T.sum(energy), state) (:408).
    W, nu, b are float32 shared variables in the reference (:398-406): rounded to float32 first."""
    W32 = _t(np.asarray(W, dtype=np.float32), dtype)
    nu32 = _t(np.asarray(nu, dtype=np.float32), dtype)
    b32 = _t(np.asarray(b, dtype=np.float32), dtype)

    def energy(x):
        rshp_b = b32.reshape(1, -1)
        rshp_nu = nu32.reshape(1, -1)
        alpha = (rshp_nu + 1.) / 2.
        energy_per_expert = alpha * torch.log(1 + ((torch.matmul(x.T, W32) + rshp_b) / rshp_nu) ** 2)
        return energy_per_expert.sum(dim=1).reshape(1, -1)

    return _with_grad(energy, X, dtype)


def funnel_literal(scale, X):
    """Funnel.build_energy_op as coded (mjhmc/misc/tf_distributions.py:157-165), including the [n] + [D-1, n]
    broadcast of tf.add."""
    def energy(x):
        e_x_0 = -((x[0, :] ** 2) / (scale ** 2))
        e_x_k = -((x[1:, :] ** 2) / torch.exp(x[0, :]))
        return (e_x_0 + e_x_k).sum(dim=0)

    return _with_grad(energy, X)


def funnel_neal(scale, X):
    """-log density of the distribution the reference documents (tf_distributions.py:143-147):
    x_0 ~ N(0, scale^2), x_k ~ N(0, e^{x_0}), additive constants dropped."""
    def energy(x):
        D = x.shape[0]
        return x[0] ** 2 / (2. * scale ** 2) + (x[1:] ** 2).sum(dim=0) / (2. * torch.exp(x[0])) + 0.5 * (D - 1) * x[0]

    return _with_grad(energy, X)


def sparse_image_code_literal(basis, patches, lmbda, cauchy, X, nbatch=None):
    """SparseImageCode.build_energy_op as coded (mjhmc/misc/tf_distributions.py:241-272): the reshape to
    [n_patches, -1, n_coeffs, 1], the tiled basis, batch_matmul, reduce_sum / reduce_mean.  For more than one
    active column the reshape interleaves the particle and coefficient axes."""
    B = _t(basis)                       # [img_size, n_coeffs]
    Y = _t(patches)                     # [n_patches, img_size]
    P, I = Y.shape
    C = B.shape[1]

    def energy(x):
        n_active = x.shape[1]
        pt = Y.reshape(P, 1, I)
        shaped_state = x.reshape(P, -1, C, 1)
        shaped_basis = B.reshape(1, 1, I, C).expand(P, nbatch or n_active, I, C)[:, :n_active]
        recon = torch.matmul(shaped_basis, shaped_state)[:, :, :, 0]
        rec_err = (0.5 * (pt - recon) ** 2).sum(dim=-1)
        rec_err = rec_err.mean(dim=0)
        if cauchy:
            pen = lmbda * torch.log(1 + x ** 2).sum(dim=0)
        else:
            pen = lmbda * torch.abs(x).sum(dim=0)
        return rec_err + pen

    return _with_grad(energy, X)


def sparse_image_code_per_column(basis, patches, lmbda, cauchy, X):
    """The reference graph evaluated one column at a time (n_active == 1, where its reshape is the identity on the
    patch-major state rows): the per-particle maths the engine implements for any number of particles."""
    X = np.asarray(X, dtype=np.float64)
    E = np.empty(X.shape[1])
    G = np.empty_like(X)
    for k in range(X.shape[1]):
        e, g = sparse_image_code_literal(basis, patches, lmbda, cauchy, X[:, k:k + 1], nbatch=1)
        E[k] = e[0]
        G[:, k:k + 1] = g
    return E, G
