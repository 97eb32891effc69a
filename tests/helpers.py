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
