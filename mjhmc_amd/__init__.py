"""mjhmc_amd -- MI355X-native engine for the particle-parallel hot path of rueberger/MJHMC.

Import surface mirrors the reference package root (mjhmc/__init__.py:6):
    from mjhmc_amd import MarkovJumpHMC, ControlHMC
    from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC
    from mjhmc_amd.misc.distributions import LambdaDistribution
"""
from mjhmc_amd.samplers.markov_jump_hmc import MarkovJumpHMC, ControlHMC  # noqa: F401

__all__ = ['MarkovJumpHMC', 'ControlHMC', 'samplers', 'misc']
