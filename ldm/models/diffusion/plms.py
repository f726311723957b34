from adaface_amd.ldm.models.diffusion.plms import PLMSSampler  # noqa: F401
