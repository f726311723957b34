from adaface_amd.ldm.models.autoencoder import AutoencoderKL, DiagonalGaussianDistribution  # noqa: F401
