"""Import-path twin of mjhmc/misc/tf_distributions.py: the reference keeps its TensorFlow-backed distributions
(TFGaussian :179-201, Funnel :142-177, SparseImageCode :204-272) in this module; here they are ordinary device energies
and live in ``distributions``."""
from .distributions import Distribution, Funnel, SparseImageCode, TFGaussian  # noqa: F401

TensorflowDistribution = Distribution   # the reference's base class of the three (:20-140): nothing TensorFlow is left of it
