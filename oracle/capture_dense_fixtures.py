#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- G2 fixtures of the energies the reference builds symbolically:
ProductOfT (Theano), Funnel and SparseImageCode (TensorFlow 0.x).  Neither package is importable, so the values
come from the torch restatement of the reference's forward graphs with AUTOGRAD gradients
(oracle/autograd_energies.py) -- independent of the hand-derived gradients in oracle/mjhmc_oracle.py.

Also converts the particle states the reference ships (initializations/*.pickle, burn-in end points of its own
samplers) into .npz inputs.  Needs /root/reference for those; the GPU box never sees the reference.

    python oracle/capture_dense_fixtures.py [--ref /root/reference]
      -> tests/golden/g2_dense.npz, tests/golden/ref_init_states.npz
"""
import argparse
import hashlib
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tests', 'golden')

from oracle import autograd_energies as ag                      # noqa: E402
from tests.helpers import ref_init_weights, sic_problem         # noqa: E402


def digest(a):
    return np.frombuffer(hashlib.sha1(np.ascontiguousarray(a, dtype=np.float64).tobytes()).digest()[:8], dtype=np.uint64)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    args = ap.parse_args()
    import torch
    out = {}

    # --- ProductOfT (distributions.py:420-433 + T.grad): 36 x 25 as search/MJHMC_poe_36 runs it, and 512 x 64 ----
    for D, n, seed in ((36, 25, 11), (512, 64, 12)):
        W, lognu = ref_init_weights(D, D)
        nu = np.exp(lognu)
        b = np.zeros(D) if D == 36 else 0.1 * np.random.RandomState(5).randn(D)
        X = np.random.RandomState(seed).randn(D, n) * 1.5
        E, g = ag.product_of_t(W, nu, b, X)
        E32, g32 = ag.product_of_t(W, nu, b, X, dtype=torch.float32)   # the reference's arithmetic: float32 graph
        tag = 'pot_%dx%d' % (D, n)
        out.update({tag + '_X': X, tag + '_E': E, tag + '_g': g, tag + '_E32': E32, tag + '_g32': g32,
                    tag + '_b': b, tag + '_Wdigest': digest(W)})
        if D == 36:
            out.update({tag + '_W': W, tag + '_lognu': lognu})

    # --- Funnel as coded (tf_distributions.py:157-165 + tf.gradients), scale 1 and 3; the documented density -------
    rs = np.random.RandomState(21)
    for scale in (1.0, 3.0):
        X = rs.randn(10, 50)
        X[0] *= 1.5
        E, g = ag.funnel_literal(scale, X)
        tag = 'funnel_lit_s%d_10x50' % int(scale)
        out.update({tag + '_X': X, tag + '_E': E, tag + '_g': g})
    for D, n in ((10, 50), (32, 64)):
        X = rs.randn(D, n)
        X[0] *= 3.0
        X[1:] *= np.exp(X[0] / 2.)
        E, g = ag.funnel_neal(3.0, X)
        tag = 'funnel_neal_s3_%dx%d' % (D, n)
        out.update({tag + '_X': X, tag + '_E': E, tag + '_g': g})

    # --- SparseImageCode (tf_distributions.py:241-272 + tf.gradients) -------------------------------------------
    # one active column: the graph as coded; several columns: the same graph column by column
    for P, n, cauchy in ((1, 1, True), (1, 8, True), (1, 8, False), (9, 1, True), (9, 4, True)):
        B, imgs, a0 = sic_problem(0, n_patches=P)
        patches = imgs[:, :P].T
        X = a0[:, None] + 0.3 * np.random.RandomState(31 + n).randn(P * 1024, n)
        if n == 1:
            E, g = ag.sparse_image_code_literal(B, patches, 0.01, cauchy, X)
        else:
            E, g = ag.sparse_image_code_per_column(B, patches, 0.01, cauchy, X)
        tag = 'sic_p%d_n%d_%s' % (P, n, 'cauchy' if cauchy else 'laplace')
        out.update({tag + '_X': X, tag + '_E': E, tag + '_g': g, tag + '_Bdigest': digest(B)})
    np.savez_compressed(os.path.join(OUT, 'g2_dense.npz'), **out)
    print('g2_dense.npz:', len(out), 'arrays')

    # --- states shipped by the reference ---------------------------------------------------------------------------
    init = {}
    with open(os.path.join(args.ref, 'initializations', 'ProductOfT_6123388416598428958.pickle'), 'rb') as f:
        pot = pickle.load(f, encoding='latin1')
    Xp = np.asarray(pot[0], dtype=np.float64)[:, :256]          # MJHMC end points, 36 x 1000 in the file
    W, lognu = ref_init_weights(36, 36)
    E, g = ag.product_of_t(W, np.exp(lognu), np.zeros(36), Xp)
    init.update({'pot36_X': Xp, 'pot36_E': E, 'pot36_g': g})
    with open(os.path.join(args.ref, 'initializations', 'SparseImageCode_-2828851975638192263.pickle'), 'rb') as f:
        sic = pickle.load(f, encoding='latin1')
    B, imgs, _ = sic_problem(0, n_patches=1)
    for name, k in (('sic_mj', 0), ('sic_ctl', 3)):                # MJHMC and control end points, 1024 x 10 each
        Xs = np.asarray(sic[k], dtype=np.float64)
        E, g = ag.sparse_image_code_per_column(B, imgs[:, :1].T, 0.01, True, Xs)
        init.update({name + '_X': Xs, name + '_E': E, name + '_g': g})
    np.savez_compressed(os.path.join(OUT, 'ref_init_states.npz'), **init)
    print('ref_init_states.npz:', {k: v.shape for k, v in init.items()})


if __name__ == '__main__':
    main()
