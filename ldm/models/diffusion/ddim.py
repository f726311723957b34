from adaface_amd.ldm.models.diffusion.ddim import DDIMSampler  # noqa: F401
