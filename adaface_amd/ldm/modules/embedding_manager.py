"""Inference subset of ldm.modules.embedding_manager.EmbeddingManager (reference embedding_manager.py:940-2259): the
hook the CLIP text tower calls between its token-embedding lookup and its encoder (encoders/modules.py:214-215).

What it does on the txt2img path (embedding_manager.py:1292-1586):
  * "tucks" the 16 U-Net cross-attention layers into the batch axis: [B, N, 768] -> [16 B, N, 768], the 16 copies of an
    instance adjacent (:1342-1353);
  * for every placeholder string present in the prompts, replaces the embeddings at its FIRST occurrence in each
    instance, and at the K-1 positions after it, by the subject's learned vectors — a [16, K, 768] tensor, vector k of
    layer l going to copy l (:1355-1366, 1501-1563).  For a non-zero-shot checkpoint those come from a
    StaticLayerwiseEmbedding (:360-537: low-rank basis combination + per-vector LayerNorm + bias);
  * records placeholder2indices (positions over the ORIGINAL batch, K consecutive per instance, :1695-1718) and
    prompt_emb_mask (everything but BOS 49406 / EOS-pad 49407, :1640-1644) — the `extra_info` the U-Net reads.

PARITY UNPINNED.  The reference module cannot be imported offline (its import of subj_basis_generator.py:22 fetches a
tokenizer at import time) and the reference holds no fixture for it; this is a restatement from the source text, tested
for the properties that text states (tests/test_host_cpu.py) and, end to end with the CLIP tower, against the oracle's
own restatement.  Not built: the zero-shot branch (SubjBasisGenerator / Arc2Face: SURVEY.md §8f-4, weight-blocked),
ada embeddings, every training-time loss and cache, cls-delta string merging.

Checkpoints.  The reference's `embeddings_gs-*.pt` pickles whole nn.Module objects (embedding_manager.py:1820-1835), which
only an executing unpickler can read; this class reads with torch.load(weights_only=True) and therefore accepts a
TENSOR-ONLY file of the same content:
    {"string_to_token": {str: int}, "token2num_vectors": {str: int}, "use_conv_attn_kernel_size": int,
     "subject_strings": [...], "background_strings": [...],
     "string_to_static_embedder": {str: tensor [16, K, D]                      (ready-made layerwise vectors), or
                                        {"basis_rand_weights": [16, K, r], "basis_comm_weights": [1, K, r],
                                         "basis_vecs": [K, r - N, D], "pre_vecs": [K, N, D] (optional), "bias": [16, K, D]}}}
and raises with that explanation when the safe loader refuses a file.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


class StaticLayerwiseEmbedding(nn.Module):
    """embedding_manager.py:360-537, the trained (non-zero-shot) form: forward() returns [16, K, D]."""

    def __init__(self, basis_rand_weights, basis_comm_weights, basis_vecs, bias, pre_vecs=None):
        super().__init__()
        self.basis_rand_weights = nn.Parameter(basis_rand_weights.float(), requires_grad=False)   # [16, K, r]
        self.basis_comm_weights = nn.Parameter(basis_comm_weights.float(), requires_grad=False)   # [1, K, r]
        self.basis_vecs = nn.Parameter(basis_vecs.float(), requires_grad=False)                   # [K, r - N, D]
        self.pre_vecs = None if pre_vecs is None else nn.Parameter(pre_vecs.float(), requires_grad=False)   # [K, N, D]
        self.bias = nn.Parameter(torch.as_tensor(bias).float(), requires_grad=False)              # [16, K, D] (or 0)
        self.num_layers, self.K = self.basis_rand_weights.shape[:2]
        self.out_emb_dim = self.basis_vecs.shape[-1]

    def forward(self, static_zs_embs=None):
        if static_zs_embs is not None:       # zero-shot (:507-514): [BS, 16, K, D] -> [(BS 16), K, D], nothing learned here
            return static_zs_embs.reshape(-1, *static_zs_embs.shape[2:])
        w = self.basis_rand_weights + self.basis_comm_weights                                     # :503
        vecs = self.basis_vecs if self.pre_vecs is None else torch.cat([self.pre_vecs, self.basis_vecs], dim=1)   # :517-521
        D = self.out_emb_dim
        # per k: [16, r] @ [r, D]; LayerNorm without affine on every (layer, k) vector; / sqrt(D); + bias   (:523-537)
        out = torch.stack([F.layer_norm(w[:, k] @ vecs[k], (D,)) for k in range(self.K)], dim=1) / math.sqrt(D)
        return out + self.bias


def extract_first_index_in_each_instance(token_indices):
    """ldm/util.py:2114-2124: of the (row, col) hits returned by torch.where, the first one of every row."""
    rows, cols = token_indices
    keep = torch.ones_like(rows, dtype=torch.bool)
    keep[1:] = rows[1:] != rows[:-1]
    return rows[keep], cols[keep]


class EmbeddingManager(nn.Module):
    def __init__(self, text_embedder=None, subject_strings=None, background_strings=None, initializer_strings=None,
                 list_initializer_word_weights=None, subj_name_to_cls_delta_string=None,
                 subj_name_to_cls_delta_word_weights=None, token2num_vectors=None, skip_loading_token2num_vectors=False,
                 use_layerwise_embedding=True, out_emb_dim=768, num_unet_ca_layers=16, layerwise_lora_rank=10,
                 layer_idx2ca_layer_idx=None, use_conv_attn_kernel_size=-1, do_zero_shot=False,
                 num_vectors_per_subj_token=1, **kwargs):
        super().__init__()
        object.__setattr__(self, "text_embedder", text_embedder)      # not a submodule: the tower owns its own weights
        # zero-shot identity path (embedding_manager.py:1406-1441, face subjects at inference): the
        # subject's 16 x K embeddings are GENERATED per call from an ArcFace embedding: arc2face_text_encoder (a
        # CLIPTextModelWrapper, "passed from ddpm.py", :1098-1100) -> one SubjBasisGenerator per placeholder.  Neither has
        # weights offline; both are built random-init and take state dicts.  PARITY UNPINNED for the manager-side plumbing.
        self.do_zero_shot = bool(do_zero_shot)
        self.string_to_subj_basis_generator_dict = nn.ModuleDict()
        self.arc2face_text_encoder = None
        self.zs_image_feat_dict = {}
        self.zs_out_id_embs_scale_range = (1.0, 1.0)
        self.zs_arc2face_inverse_prompt_embs_inf_type = kwargs.get("zs_arc2face_inverse_prompt_embs_inf_type", "full_half_pad")
        self.zs_tokenizer = kwargs.get("tokenizer", None)
        self.zs_arc2face_input_ids = kwargs.get("zs_arc2face_input_ids", None)     # ids of "photo of a id person" [1, 77]
        self.zs_arcface_token_id = kwargs.get("zs_arcface_token_id", None)
        self.use_layerwise_embedding = use_layerwise_embedding
        self.num_unet_ca_layers = num_unet_ca_layers
        self.num_layers_per_embedder = num_unet_ca_layers if use_layerwise_embedding else 1
        self.out_emb_dim = out_emb_dim
        self.subject_strings = list(subject_strings or [])
        self.background_strings = list(background_strings or [])
        self.background_string_dict = {s: True for s in self.background_strings}
        self.subject_string_dict = {s: True for s in self.subject_strings}
        self.placeholder_strings = self.subject_strings + self.background_strings
        self.token2num_vectors = dict(token2num_vectors or {})
        for s in self.placeholder_strings:
            self.token2num_vectors.setdefault(s, num_vectors_per_subj_token)
        self.string_to_token_dict: "OrderedDict[str, int]" = OrderedDict()
        self.string_to_static_embedder_dict = nn.ModuleDict()
        self._static_tensors: Dict[str, torch.Tensor] = {}
        self.extended_token_embeddings = None
        self.curr_subj_is_face = False
        self.use_conv_attn_kernel_size = use_conv_attn_kernel_size
        self.iter_type = None
        self.clear_prompt_adhoc_info()

    # ---- bookkeeping the caller touches (stable_txt2img.py:402-432, ddpm.py:1003, 1058-1061) ----
    def clear_prompt_adhoc_info(self):                      # :1646-1649
        self.placeholder2indices = {}
        self.img_mask = None
        self.prompt_emb_mask = None

    def set_curr_iter_type(self, embman_iter_type):         # :1689-1693
        self.iter_type = embman_iter_type

    def set_conv_attn_kernel_size(self, use_conv_attn_kernel_size=-1):   # :1759-1775
        self.use_conv_attn_kernel_size = -1 if use_conv_attn_kernel_size is None else use_conv_attn_kernel_size

    def set_zs_image_features(self, zs_clip_features, zs_id_embs, zs_out_id_embs_scale_range=(1.0, 1.0),
                              add_noise_to_zs_id_embs=False):                      # :1786-1808 (inference: no noise)
        if zs_clip_features is not None:
            subj_feat, bg_feat = zs_clip_features.chunk(2, dim=1)
        else:
            subj_feat = bg_feat = None
        self.zs_image_feat_dict = {'subj': subj_feat, 'bg': bg_feat, 'id': zs_id_embs}
        self.zs_out_id_embs_scale_range = zs_out_id_embs_scale_range

    def add_zero_shot_placeholder(self, string: str, token: int, num_vectors: int = 16, clip_config=None,
                                  inverse_prompt_input_ids=None, pad_token_id=49407):
        """A face subject whose embeddings come from a SubjBasisGenerator (:1163-1177)."""
        from adaface_amd.ldm.modules.subj_basis_generator import SubjBasisGenerator
        if not self.do_zero_shot:
            raise RuntimeError("add_zero_shot_placeholder on an EmbeddingManager built with do_zero_shot=False")
        self.string_to_token_dict[string] = int(token)
        self.token2num_vectors[string] = num_vectors
        if string not in self.placeholder_strings:
            self.subject_strings.append(string)
            self.subject_string_dict[string] = True
            self.placeholder_strings = self.subject_strings + self.background_strings
        self.string_to_subj_basis_generator_dict[string] = SubjBasisGenerator(
            num_out_embs_per_layer=num_vectors, num_out_layers=self.num_unet_ca_layers, output_dim=self.out_emb_dim,
            clip_config=clip_config, tokenizer=self.zs_tokenizer, inverse_prompt_input_ids=inverse_prompt_input_ids,
            pad_token_id=pad_token_id)
        self.curr_subj_is_face = True

    def _zero_shot_embedding(self, string, device):
        """:1406-1441: ArcFace [BS, 512] -> Arc2Face core identity embeddings -> SubjBasisGenerator -> [(BS 16), K, D]."""
        from adaface_amd.ldm.util import arc2face_forward_face_embs
        if string in self.background_string_dict:
            raise NotImplementedError("zero-shot background placeholders are not built (SURVEY.md §8f-4)")
        if self.arc2face_text_encoder is None:
            raise RuntimeError("zero-shot: set embedding_manager.arc2face_text_encoder (a CLIPTextModelWrapper with the Arc2Face "
                               "encoder's weights; the reference downloads 'arc2face/models', :1777-1784)")
        zs_id_embs = self.zs_image_feat_dict.get('id')
        if zs_id_embs is None:
            raise RuntimeError("zero-shot: call set_zs_image_features(zs_clip_features, zs_id_embs) first")
        zs_id_embs = zs_id_embs.to(device)
        with torch.no_grad():
            _, arc2face_id_embs = arc2face_forward_face_embs(self.zs_tokenizer, self.arc2face_text_encoder, zs_id_embs,
                                                             return_full_and_core_embs=True,
                                                             input_ids=self.zs_arc2face_input_ids,
                                                             arcface_token_id=self.zs_arcface_token_id)
            gen = self.string_to_subj_basis_generator_dict[string]
            static_zs_embs, inverse_embs = gen(self.zs_image_feat_dict.get('subj'), zs_id_embs, arc2face_id_embs,
                                               self.zs_out_id_embs_scale_range[0], is_face=self.curr_subj_is_face,
                                               is_training=False,
                                               arc2face_inverse_prompt_embs_inf_type=self.zs_arc2face_inverse_prompt_embs_inf_type)
        self.arc2face_inverse_prompt_embs = inverse_embs
        return static_zs_embs.reshape(-1, *static_zs_embs.shape[2:])              # StaticLayerwiseEmbedding.forward(:509)

    def extend_placeholders(self, new_subject_strings, new_background_strings, num_vectors_per_subj_token,
                            num_vectors_per_bg_token):
        """:1588-1632.  New placeholder strings need new rows in the CLIP token table (extend_clip_text_embedder), which
        needs the tokenizer; strings already known are no-ops, as in the reference."""
        for k in new_subject_strings or []:
            if k is None or k in self.subject_strings:
                continue
            self.subject_strings.append(k)
            self.subject_string_dict[k] = True
            self.placeholder_strings.append(k)
            self.token2num_vectors[k] = num_vectors_per_subj_token
        for k in new_background_strings or []:
            if k is None or k in self.background_strings:
                continue
            self.background_strings.append(k)
            self.background_string_dict[k] = True
            self.placeholder_strings.append(k)
            self.token2num_vectors[k] = num_vectors_per_bg_token

    def add_placeholder(self, string: str, token: int, embedder, is_bg: bool = False):
        """Register one placeholder: its token id and either a [16, K, D] tensor or a StaticLayerwiseEmbedding."""
        if is_bg and string not in self.background_strings:
            self.background_strings.append(string)
            self.background_string_dict[string] = True
        if not is_bg and string not in self.subject_strings:
            self.subject_strings.append(string)
            self.subject_string_dict[string] = True
        if string not in self.placeholder_strings:
            self.placeholder_strings.append(string)
        self.string_to_token_dict[string] = int(token)
        if isinstance(embedder, nn.Module):
            self.string_to_static_embedder_dict[string] = embedder
            self.token2num_vectors[string] = embedder.K
        else:
            t = torch.as_tensor(embedder).float()
            if t.dim() != 3 or t.shape[0] != self.num_layers_per_embedder:
                raise ValueError(f"static embedding must be [{self.num_layers_per_embedder}, K, D], got {tuple(t.shape)}")
            self._static_tensors[string] = t
            self.token2num_vectors[string] = t.shape[1]

    def load(self, ckpt_paths, src_placeholders=None, extend_prompt2token_proj_attention_multiplier=-1,
             load_old_embman_ckpt=False):
        """:1840-2052, for tensor-only files (see the module docstring)."""
        if isinstance(ckpt_paths, str):
            ckpt_paths = [ckpt_paths]
        for path in ckpt_paths:
            path = path.split(":")[0]
            try:
                ckpt = torch.load(path, map_location="cpu", weights_only=True)
            except Exception as e:
                raise RuntimeError(
                    f"{path}: the safe loader (torch.load(weights_only=True)) refused this file ({type(e).__name__}). "
                    "The reference's embedding checkpoints pickle nn.Module objects; convert to the tensor-only layout "
                    "described in adaface_amd/ldm/modules/embedding_manager.py") from e
            self.set_conv_attn_kernel_size(ckpt.get("use_conv_attn_kernel_size", None))
            bg = set(ckpt.get("background_strings", []))
            for s, tok in ckpt["string_to_token"].items():
                emb = ckpt["string_to_static_embedder"][s]
                if isinstance(emb, dict):
                    emb = StaticLayerwiseEmbedding(emb["basis_rand_weights"], emb["basis_comm_weights"], emb["basis_vecs"],
                                                   emb.get("bias", 0.0), emb.get("pre_vecs", None))
                self.add_placeholder(s, int(tok), emb, is_bg=s in bg)
            for s, n in ckpt.get("token2num_vectors", {}).items():
                self.token2num_vectors[s] = int(n)

    # ---- the hook ----
    @torch.no_grad()
    def forward(self, tokenized_text, embedded_text):
        """tokenized_text [B, N] int64, embedded_text [B, N, D] -> [16 B, N, D] (or [B, N, D] when not layerwise)."""
        B, N = tokenized_text.shape
        device = tokenized_text.device
        self.clear_prompt_adhoc_info()
        L = self.num_unet_ca_layers
        embedded_text = embedded_text.clone()
        tok = tokenized_text
        if self.use_layerwise_embedding:                                                     # :1342-1353
            embedded_text = embedded_text.unsqueeze(1).repeat(1, L, 1, 1).view(B * L, N, -1)
            tok = tokenized_text.unsqueeze(1).repeat(1, L, 1).view(B * L, N)
        for string, token in self.string_to_token_dict.items():                             # :1355
            idx = torch.where(tok == token)
            if idx[0].numel() == 0:
                continue
            rows, cols = extract_first_index_in_each_instance(idx)                           # :1366
            occurs = rows.numel() // self.num_layers_per_embedder                            # :1382
            if string in self.string_to_subj_basis_generator_dict:                          # :1406-1441 (zero-shot)
                subj = self._zero_shot_embedding(string, device)
                if subj.shape[0] < L * occurs and (L * occurs) % subj.shape[0] == 0 and subj.shape[0] != L:
                    subj = subj.repeat((L * occurs) // subj.shape[0], 1, 1)                  # :1447-1449
            elif string in self.string_to_static_embedder_dict:
                subj = self.string_to_static_embedder_dict[string].to(device)(None)          # :1396, 1501
            else:
                subj = self._static_tensors[string].to(device)                               # :1504-1505
            subj = subj.to(embedded_text.dtype)
            K = self.token2num_vectors[string]
            if cols.max().item() + K > N:
                raise ValueError(f"placeholder '{string}' with {K} vectors runs past the end of the prompt")
            for k in range(K):                                                               # :1509-1563
                e_k = subj[:, k]
                if e_k.shape[0] == L:
                    e_k = e_k.repeat(occurs, 1)
                elif e_k.shape[0] != L * occurs:                                             # :1548-1549
                    raise ValueError(f"placeholder '{string}': {e_k.shape[0]} embedding rows for {occurs} occurrences x {L} layers")
                embedded_text[(rows, cols + k)] = e_k
            self.update_placeholder_indices(tokenized_text, string, token, K)
        self.update_prompt_masks(tokenized_text)                                             # :1324
        return embedded_text

    def update_placeholder_indices(self, tokenized_text, placeholder_string, placeholder_token, num_vectors_per_subj_token,
                                   placeholder_is_bg=False):
        """:1695-1718."""
        idx = torch.where(tokenized_text == placeholder_token)
        if idx[0].numel() == 0:
            self.placeholder2indices[placeholder_string] = None
            return
        idx_b, idx_n = extract_first_index_in_each_instance(idx)
        K = num_vectors_per_subj_token
        if K > 1:
            bs = idx_b.shape[0]
            idx_b = idx_b.unsqueeze(1).repeat(1, K).view(-1)
            idx_n = idx_n.unsqueeze(1).repeat(1, K).view(-1) + torch.arange(K, device=tokenized_text.device).repeat(bs)
        self.placeholder2indices[placeholder_string] = (idx_b, idx_n)

    def update_prompt_masks(self, tokenized_text, tokenized_text_repeated=False):
        """:1640-1644: everything but the start (49406) and end / padding (49407) tokens."""
        self.prompt_emb_mask = ((tokenized_text != 49406) & (tokenized_text != 49407)).float().unsqueeze(2)
