"""Surface of ldm.modules.embedding_manager.EmbeddingManager that scripts/stable_txt2img.py touches
(reference stable_txt2img.py:402-432, ddpm.py:1058-1061).  The real class (embedding_manager.py:940-2259)
is the conditioning PRODUCER — it runs once per prompt, needs the CLIP tokenizer/weights and pickled
nn.Module checkpoints, and is out of scope for the denoising path.  This stub
keeps the attribute/method names so the caller's plumbing runs; anything that would need the real
arithmetic raises.
"""
from __future__ import annotations

import torch.nn as nn


class EmbeddingManager(nn.Module):
    def __init__(self, *args, subject_strings=None, background_strings=None, num_vectors_per_subj_token=1,
                 use_layerwise_embedding=True, **kwargs):
        super().__init__()
        self.subject_strings = list(subject_strings or [])
        self.background_strings = list(background_strings or [])
        self.token2num_vectors = {s: num_vectors_per_subj_token for s in self.subject_strings}
        self.extended_token_embeddings = None
        self.curr_subj_is_face = False
        self.do_zero_shot = False
        self.use_conv_attn_kernel_size = -1
        self.placeholder2indices = {}
        self.prompt_emb_mask = None
        self.use_layerwise_embedding = use_layerwise_embedding

    def extend_placeholders(self, subj, bg, n_subj, n_bg):
        for s in subj or []:
            self.token2num_vectors[s] = n_subj
        for s in bg or []:
            self.token2num_vectors[s] = n_bg

    def load(self, paths, load_old_embman_ckpt=False):
        raise NotImplementedError(
            "embedding checkpoints are pickled nn.Module objects of the reference's classes "
            "(embedding_manager.py:1820-1868); loading them is out of scope for the denoising path")

    def forward(self, *args, **kwargs):
        raise NotImplementedError("EmbeddingManager.forward (token substitution inside CLIP) is out of scope; "
                                  "feed LatentDiffusion.get_learned_conditioning a pre-computed embedding")
