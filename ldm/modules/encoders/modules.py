from adaface_amd.ldm.modules.encoders.modules import FrozenCLIPEmbedder  # noqa: F401
