from adaface_amd.ldm.modules.diffusionmodules.util import *  # noqa: F401,F403
