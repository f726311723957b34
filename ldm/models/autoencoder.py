from adaface_amd.ldm.models.autoencoder import AutoencoderKL  # noqa: F401
