from adaface_amd.ldm.models.diffusion.ddpm import DDPM, DiffusionWrapper, LatentDiffusion  # noqa: F401
