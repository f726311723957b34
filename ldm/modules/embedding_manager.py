from adaface_amd.ldm.modules.embedding_manager import EmbeddingManager  # noqa: F401
