from adaface_amd.ldm.modules.arc2face_models import CLIPTextModelWrapper  # noqa: F401
