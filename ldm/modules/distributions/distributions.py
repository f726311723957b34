"""ldm.modules.distributions.distributions at the reference's dotted path (distributions.py:24-62)."""
from adaface_amd.ldm.models.autoencoder import DiagonalGaussianDistribution  # noqa: F401
