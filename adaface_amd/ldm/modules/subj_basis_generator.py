"""Drop-in for ldm.modules.subj_basis_generator.SubjBasisGenerator (reference subj_basis_generator.py:369-622), the
INFERENCE / FACE branch: the generator that turns Arc2Face's 16 core identity embeddings into the [BS, 16 layers, K, 768]
subject embeddings the EmbeddingManager puts into the prompt.

    arc2face_id_embs [BS, 16, 768]
      -> arc2face_inverse_face_prompt_embs(prompt2token_proj, ..., hidden_state_layer_weights [1, 2, 4])   (:513-523)
      -> core_id_embs [BS, 16, 768] repeated over the 16 layers                                            (:548-549)
      -> * out_id_embs_scale + pad_embeddings[2 : 2 + K] * (1 - out_id_embs_scale)                         (:553-554)

prompt2token_proj is a CLIPTextModelWrapper (the HIP CLIP tower).  PARITY UNPINNED for this class: the reference module
fetches a tokenizer at import time (subj_basis_generator.py:22) and its weights (Arc2Face, AdaFace zero-shot checkpoints)
do not exist offline; the two tower drives it is made of are pinned (tests/golden/golden_clip.npz zs_*).
Not built: the background branch (CrossAttention prompt translator over CLIP image features, :536-546), the object
branch (DINO features through ExpandEmbs, :524-528), training (gradient scalers, attention extension).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from adaface_amd.ldm.modules.arc2face_models import CLIPTextModelWrapper
from adaface_amd.ldm.util import arc2face_inverse_face_prompt_embs


class SubjBasisGenerator(nn.Module):
    def __init__(self, num_heads=6, num_id_vecs={'subj': 77, 'bg': 257}, num_out_embs_per_layer=4, num_out_layers=16,
                 image_embedding_dim=768, dino_embedding_dim=384, output_dim=768, placeholder_is_bg: bool = False,
                 prompt2token_proj_grad_scale: float = 0.4, zs_extra_words_scale: float = 0.5,
                 learnable_hidden_state_weights_scheme: str = 'per-layer', bg_prompt_translator_has_to_out_proj: bool = False,
                 clip_config=None, tokenizer=None, inverse_prompt_input_ids=None, pad_token_id=49407):
        super().__init__()
        if placeholder_is_bg:
            raise NotImplementedError("SubjBasisGenerator: the background branch (prompt translator over CLIP image "
                                      "features) is not built (SURVEY.md §8f-4 covers the face path)")
        self.placeholder_is_bg = False
        self.num_out_layers = num_out_layers
        self.num_out_embs_per_layer = num_out_embs_per_layer
        self.num_out_embs = num_out_layers * num_out_embs_per_layer
        self.output_dim = output_dim
        self.num_id_vecs = num_id_vecs['subj']
        self.zs_extra_words_scale = zs_extra_words_scale
        self.output_scale = output_dim ** -0.5
        self.pos_embs = nn.Parameter(torch.randn(1, self.num_id_vecs, output_dim), requires_grad=False)   # unused on this branch
        self.pos_embs_ln = nn.LayerNorm(output_dim)
        self.prompt2token_proj = CLIPTextModelWrapper(clip_config)
        self.prompt2token_proj_grad_scale = prompt2token_proj_grad_scale
        self.prompt2token_proj_attention_multiplier = -1
        # initialize_hidden_state_layer_weights (:562-581): last three layers, [1, 2, 4]
        if learnable_hidden_state_weights_scheme == 'none':
            self.hidden_state_layer_weights = None
        elif learnable_hidden_state_weights_scheme == 'per-layer':
            self.hidden_state_layer_weights = nn.Parameter(torch.tensor([[1.0], [2.0], [4.0]]), requires_grad=False)
        else:
            raise ValueError(learnable_hidden_state_weights_scheme)
        self.pad_embeddings = None
        # no CLIP vocabulary offline: the tokenizer is injected, or the fixed template's ids / pad token id are given
        self.tokenizer = tokenizer
        self.inverse_prompt_input_ids = inverse_prompt_input_ids
        self.pad_token_id = pad_token_id if tokenizer is None else tokenizer.pad_token_id

    def generate_pad_embeddings(self, device):
        """:583-596: CLIPTextEmbeddings (token + position) of 77 pad tokens, [77, 768], detached."""
        emb = self.prompt2token_proj
        n = emb.clip_config["max_pos"]
        pad_tokens = torch.full((1, n), self.pad_token_id, dtype=torch.long, device=device)
        tok = emb(input_ids=pad_tokens, return_token_embs=True)[0]
        pos = emb.text_model.embeddings.position_embedding.weight.detach().to(device).float()
        self.pad_embeddings = (tok + pos[:n]).detach()

    def extend_prompt2token_proj_attention(self, *a, **k):
        raise NotImplementedError("prompt2token_proj attention extension (CLIPAttentionMKV) is a training-time feature")

    def forward(self, clip_features, raw_id_embs, arc2face_id_embs, out_id_embs_scale, is_face, is_training,
                arc2face_inverse_prompt_embs_inf_type='full_half_pad'):
        if not is_face:
            raise NotImplementedError("SubjBasisGenerator: only the face branch (is_face=True) is built")
        if is_training:
            raise NotImplementedError("SubjBasisGenerator: inference only")
        assert arc2face_id_embs is not None
        dev = arc2face_id_embs.device
        if self.pad_embeddings is None:
            self.generate_pad_embeddings(dev)
        else:
            self.pad_embeddings = self.pad_embeddings.to(dev)
        inverse_embs, core_id_embs = arc2face_inverse_face_prompt_embs(
            self.tokenizer, self.prompt2token_proj, arc2face_id_embs, list_extra_words=None,
            return_emb_types=[arc2face_inverse_prompt_embs_inf_type, 'core'], pad_embeddings=self.pad_embeddings,
            hidden_state_layer_weights=self.hidden_state_layer_weights, input_max_length=77,
            zs_extra_words_scale=self.zs_extra_words_scale, input_ids=self.inverse_prompt_input_ids)
        # [BS, 16, 768] -> [BS, 16 layers, 16, 768]
        output_embs = core_id_embs.unsqueeze(1).repeat(1, self.num_out_layers, 1, 1)
        K = self.num_out_embs_per_layer
        output_embs = output_embs * out_id_embs_scale + self.pad_embeddings[2:2 + K].unsqueeze(0) * (1 - out_id_embs_scale)
        return output_embs, inverse_embs
