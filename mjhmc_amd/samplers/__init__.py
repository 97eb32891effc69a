"""Sampler classes of the drop-in (mirrors mjhmc.samplers)."""
