from adaface_amd.ldm.modules.diffusionmodules.openaimodel import UNetModel  # noqa: F401
