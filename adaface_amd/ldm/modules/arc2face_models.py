"""Drop-in for ldm.modules.arc2face_models.CLIPTextModelWrapper (reference arc2face_models.py:175-280): a CLIP text
tower that accepts precomputed token embeddings (`input_token_embs`), can return the token embeddings of its input ids
(`return_token_embs=True`) and can blend its last hidden states (`hidden_state_layer_weights`, normalised to sum 1,
:230-243) before the final LayerNorm — the two text encoders of the zero-shot identity path (the
Arc2Face text encoder and SubjBasisGenerator.prompt2token_proj).  State-dict keys are transformers' (`text_model.*`).

The arithmetic is the HIP CLIP tower of the conditioning producer (af_clip_embed_tokens / af_clip_text_forward3).
Not built: the attention extension `extend_clip_attention_MKV_multiplier` (arc2face_models.py:283-302, CLIPAttentionMKV
:16-173: a training-time widening of the key / value projections): checkpoints saved with an extended prompt2token_proj
are refused at load time by their tensor shapes.
"""
from __future__ import annotations

from typing import Optional

import torch

from adaface_amd import layout
from adaface_amd.ldm._hipmodule import HipModule, build_param_tree
from adaface_amd.ldm.modules.encoders.modules import CLIP_VIT_L14_TEXT


class CLIPTextModelWrapper(HipModule):
    _ckpt_prefix = "cond_stage_model.transformer."

    def __init__(self, clip_config: Optional[dict] = None, eos_token_id: int = 2):
        super().__init__()
        self.clip_config = dict(CLIP_VIT_L14_TEXT if clip_config is None else clip_config)
        self.eos_token_id = eos_token_id      # text_model.eos_token_id (arc2face_models.py:250)
        build_param_tree(self, {"text_model." + k: v for k, v in layout.clip_text_param_shapes(**self.clip_config).items()})

    @classmethod
    def from_pretrained(cls, *a, **k):
        raise RuntimeError("CLIPTextModelWrapper.from_pretrained: no model files offline; construct it and load a "
                           "state_dict with transformers' CLIPTextModel keys (text_model.*)")

    def _engine_kwargs(self):
        return {"clip": dict(self.clip_config)}

    @property
    def dtype(self):
        return torch.float32

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, input_token_embs=None, hidden_state_layer_weights=None,
                return_token_embs=False):
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        if attention_mask is not None or position_ids is not None or output_attentions or output_hidden_states:
            raise NotImplementedError("CLIPTextModelWrapper: attention_mask / position_ids / attention or hidden-state outputs "
                                      "are not used on the zero-shot identity path")
        ids = input_ids.view(-1, input_ids.shape[-1])
        eng = self.engine(ids.device)
        if return_token_embs:                                   # arc2face_models.py:192-193
            return eng.clip_embed_tokens(ids)
        emb = eng.clip_embed_tokens(ids) if input_token_embs is None else input_token_embs
        if hidden_state_layer_weights is None:                  # :228-229: the last hidden state
            w = [0.0, 0.0, 1.0]
        else:                                                   # :231-243: weights over the last n states, summing to 1
            hw = torch.as_tensor(hidden_state_layer_weights, dtype=torch.float64).detach().cpu()
            if hw.dim() == 2 and hw.shape[1] != 1:
                raise NotImplementedError("per-channel hidden_state_layer_weights ([n, 768]) are a training-time option")
            hw = hw.reshape(-1)
            if not 1 <= hw.numel() <= 3:
                raise NotImplementedError(f"{hw.numel()} blended hidden states (the reference uses 3)")
            hw = hw / hw.sum()
            w = [0.0] * (3 - hw.numel()) + [float(v) for v in hw]
        last = eng.clip_text_forward3(emb.to(ids.device), w[0], w[1], w[2])
        # pooled output (arc2face_models.py:250-269): the state at the end-of-text position
        if self.eos_token_id == 2:
            pos = ids.to(torch.int).argmax(dim=-1)
        else:
            pos = (ids.to(torch.int) == self.eos_token_id).int().argmax(dim=-1)
        pooled = last[torch.arange(last.shape[0], device=last.device), pos]
        return (last, pooled)
