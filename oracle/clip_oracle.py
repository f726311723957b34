"""CPU ORACLE (test infrastructure, NOT product code) for the conditioning PRODUCER in front of the denoising path
(SURVEY.md §8f-2): the CLIP text tower as the reference drives it, and the inference subset of
EmbeddingManager.forward that patches the token embeddings in between.

Only tests/ (and tests/golden/gen_golden_clip.py) may import this module; the product never does.

Pinning
  * clip_text_forward — the transformer arithmetic lives in the third-party `transformers` package, which the
    reference pins only loosely (environment.yaml: transformers==4.25.1; importable here: 5.15).  The reference's own
    FrozenCLIPEmbedder cannot be constructed offline (encoders/modules.py:184-185 call from_pretrained), so this
    restatement is pinned against a RANDOMLY INITIALISED transformers.CLIPTextModel built from a config, driven the way
    the reference's patched forwards drive it (tests/golden/gen_golden_clip.py -> tests/golden/golden_clip.npz,
    tests/test_oracle_golden.py).
  * zero-shot identity path: arc2face_forward_face_embs / arc2face_inverse_face_prompt_embs are the
    reference's two drives of CLIPTextModelWrapper; their transformer arithmetic is pinned like clip_text_forward (plain
    last state, and the last three states weighted [1, 2, 4] / 7; golden_clip.npz zs_*); subj_basis_generator_face and
    the way the EmbeddingManager consumes it are restatements from the source text: PARITY UNPINNED (the module cannot be
    imported offline and the real Arc2Face / AdaFace weights do not exist here).
  * embedding_manager_patch / static_layerwise_embedding — PARITY UNPINNED: ldm/modules/embedding_manager.py cannot be
    imported offline (its import of subj_basis_generator.py:22 fetches a tokenizer at import time) and the reference
    holds no fixture for it.  These are restatements from the source text, checked only for the properties the text
    states (tests/test_host_cpu.py).

Every function cites the reference file:line it restates.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]
CLIP_PREFIX = "cond_stage_model.transformer.text_model."


@dataclass
class ClipConfig:
    """openai/clip-vit-large-patch14 text tower (encoders/modules.py:181: version default)."""
    vocab: int = 49408
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    intermediate: int = 3072
    max_pos: int = 77
    eps: float = 1e-5


SD15_CLIP = ClipConfig()
TINY_CLIP = ClipConfig(vocab=1000, hidden=64, layers=3, heads=4, intermediate=128)


def clip_param_shapes(cfg: ClipConfig, prefix: str = CLIP_PREFIX) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys of CLIPTextModel.text_model as SD checkpoints carry them under cond_stage_model.transformer."""
    D, Fm = cfg.hidden, cfg.intermediate
    out = {prefix + "embeddings.token_embedding.weight": (cfg.vocab, D),
           prefix + "embeddings.position_embedding.weight": (cfg.max_pos, D)}
    for i in range(cfg.layers):
        p = f"{prefix}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            out[p + f"self_attn.{n}.weight"] = (D, D)
            out[p + f"self_attn.{n}.bias"] = (D,)
        for n in ("layer_norm1", "layer_norm2"):
            out[p + n + ".weight"] = (D,)
            out[p + n + ".bias"] = (D,)
        out[p + "mlp.fc1.weight"] = (Fm, D)
        out[p + "mlp.fc1.bias"] = (Fm,)
        out[p + "mlp.fc2.weight"] = (D, Fm)
        out[p + "mlp.fc2.bias"] = (D,)
    out[prefix + "final_layer_norm.weight"] = (D,)
    out[prefix + "final_layer_norm.bias"] = (D,)
    return out


def clip_embed_tokens(sd: SD, ids: Tensor, prefix: str = CLIP_PREFIX) -> Tensor:
    """embeddings_forward, first step (encoders/modules.py:207-208): inputs_embeds = token_embedding(input_ids)."""
    return sd[prefix + "embeddings.token_embedding.weight"][ids]


def clip_text_forward(sd: SD, cfg: ClipConfig, inputs_embeds: Tensor, skip_weights: Sequence[float] = (0.5, 0.5),
                      prefix: str = CLIP_PREFIX) -> Tensor:
    """text_model_forward after the embedding lookup (encoders/modules.py:299-371).

    inputs_embeds [Bn, T, D]: token embeddings, already patched by the EmbeddingManager when there is one.
      + position_embedding[:T]                                             (modules.py:217-225)
      causal mask, L x CLIPEncoderLayer (transformers modeling_clip: x += out_proj(softmax(q k^T / sqrt(dh) + mask) v),
        q/k/v/out_proj with bias, pre-LN; x += fc2(quick_gelu(fc1(LN2 x))), quick_gelu(x) = x * sigmoid(1.702 x))
      encoder_states = inputs of every layer + the final output            (modules.py:262-283)
      weighted sum of the last len(skip_weights) states, weights normalised to sum 1, last element = last layer
        (modules.py:361-368, 399-403), then final_layer_norm              (modules.py:370)."""
    Bn, T, D = inputs_embeds.shape
    H, dh = cfg.heads, cfg.hidden // cfg.heads
    x = inputs_embeds + sd[prefix + "embeddings.position_embedding.weight"][:T]
    mask = torch.full((T, T), float("-inf")).triu(1)
    states = []
    for i in range(cfg.layers):
        states.append(x)
        p = f"{prefix}encoder.layers.{i}."
        n = F.layer_norm(x, (D,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], cfg.eps)
        q = F.linear(n, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]).view(Bn, T, H, dh).transpose(1, 2)
        k = F.linear(n, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"]).view(Bn, T, H, dh).transpose(1, 2)
        v = F.linear(n, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"]).view(Bn, T, H, dh).transpose(1, 2)
        att = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5 + mask, dim=-1) @ v
        att = att.transpose(1, 2).reshape(Bn, T, D)
        x = x + F.linear(att, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        n = F.layer_norm(x, (D,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], cfg.eps)
        f = F.linear(n, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        f = f * torch.sigmoid(1.702 * f)
        x = x + F.linear(f, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    states.append(x)
    if skip_weights is not None:
        w = torch.tensor(list(skip_weights), dtype=x.dtype)
        w = w / w.sum()
        x = sum(wi * st for wi, st in zip(w, states[-len(w):]))
    return F.layer_norm(x, (D,), sd[prefix + "final_layer_norm.weight"], sd[prefix + "final_layer_norm.bias"], cfg.eps)


# ----------------------------------------------------------------------------------------
# EmbeddingManager, inference subset (PARITY UNPINNED, see the module docstring)
# ----------------------------------------------------------------------------------------
def static_layerwise_embedding(basis_rand_weights: Tensor, basis_comm_weights: Tensor, basis_vecs: Tensor,
                               pre_vecs: Optional[Tensor], bias) -> Tensor:
    """StaticLayerwiseEmbedding.forward, the non-zero-shot branch (embedding_manager.py:500-537):
    weights [16, K, r] = rand + comm; per k: [16, r] @ cat(pre_vecs, basis_vecs)[k] [r, D]; LayerNorm without affine over D;
    / sqrt(D); + bias [16, K, D]."""
    w = basis_rand_weights + basis_comm_weights
    vecs = basis_vecs if pre_vecs is None else torch.cat([pre_vecs, basis_vecs], dim=1)
    K, D = vecs.shape[0], vecs.shape[-1]
    out = torch.stack([F.layer_norm(w[:, k] @ vecs[k], (D,)) for k in range(K)], dim=1) / math.sqrt(D)
    return out + bias


def first_index_in_each_instance(rows: Tensor, cols: Tensor):
    """extract_first_index_in_each_instance (ldm/util.py:2114-2124): first (row, col) hit of every row that has one."""
    keep = torch.ones_like(rows, dtype=torch.bool)
    keep[1:] = rows[1:] != rows[:-1]
    return rows[keep], cols[keep]


def embedding_manager_patch(ids: Tensor, emb: Tensor, token: int, subj_emb: Tensor, n_layers: int = 16):
    """EmbeddingManager.forward -> get_static_embedding for ONE placeholder token at inference
    (embedding_manager.py:1292-1353, 1501-1563):
      emb [B, N, D] -> unsqueeze(1).repeat(1, 16, 1, 1).view(16 B, N, D): the 16 layer copies of an instance are adjacent;
      ids likewise; first occurrence of `token` in every (repeated) row; subj_emb [16, K, D]: vector k of layer l replaces
      position first + k of copy l of every instance that has the token (subj_emb[:, k].repeat(REAL_OCCURS, 1)).
    Returns (patched emb [16 B, N, D], placeholder indices over the ORIGINAL batch as update_placeholder_indices builds
    them, embedding_manager.py:1695-1718, prompt_emb_mask [B, N, 1], :1640-1644)."""
    B, N, D = emb.shape
    K = subj_emb.shape[1]
    e = emb.unsqueeze(1).repeat(1, n_layers, 1, 1).view(B * n_layers, N, D).clone()
    t = ids.unsqueeze(1).repeat(1, n_layers, 1).view(B * n_layers, N)
    rows, cols = torch.where(t == token)
    if rows.numel():
        r1, c1 = first_index_in_each_instance(rows, cols)
        occurs = r1.numel() // n_layers
        for k in range(K):
            e[(r1, c1 + k)] = subj_emb[:, k].repeat(occurs, 1)
    rb, cb = torch.where(ids == token)
    if rb.numel():
        rb, cb = first_index_in_each_instance(rb, cb)
        idx_b = rb.unsqueeze(1).repeat(1, K).view(-1)
        idx_n = cb.unsqueeze(1).repeat(1, K).view(-1) + torch.arange(K).repeat(rb.numel())
        placeholder = (idx_b, idx_n)
    else:
        placeholder = None
    mask = ((ids != 49406) & (ids != 49407)).float().unsqueeze(2)
    return e, placeholder, mask


# ----------------------------------------------------------------------------------------
# zero-shot identity path
# ----------------------------------------------------------------------------------------
def arc2face_forward_face_embs(sd: SD, cfg: ClipConfig, input_ids: Tensor, id_token: int, face_embs: Tensor,
                               prefix: str = CLIP_PREFIX):
    """arc2face_forward_face_embs (ldm/util.py:1085-1131) on the Arc2Face text encoder (a CLIPTextModelWrapper,
    arc2face_models.py:175-280): token embeddings of "photo of a id person" (input_ids [N, 77], the tokenizer's job), the
    'id' token's embedding replaced by the ArcFace vector zero-padded to the hidden size (util.py:1111-1113), plain CLIP
    forward (last hidden state -> final LayerNorm), full [N, 77, D] and core = tokens 4:20 (util.py:1126-1128)."""
    tok = clip_embed_tokens(sd, input_ids, prefix).clone()
    tok[input_ids == id_token] = F.pad(face_embs, (0, cfg.hidden - face_embs.shape[-1]))
    full = clip_text_forward(sd, cfg, tok, skip_weights=(1.0,), prefix=prefix)
    return full, full[:, 4:20]


def clip_pad_embeddings(sd: SD, cfg: ClipConfig, pad_token: int, prefix: str = CLIP_PREFIX) -> Tensor:
    """SubjBasisGenerator.generate_pad_embeddings (subj_basis_generator.py:583-596): CLIPTextEmbeddings of 77 pad tokens =
    token_embedding[pad] + position_embedding: [77, D]."""
    return sd[prefix + "embeddings.token_embedding.weight"][pad_token].unsqueeze(0) + \
        sd[prefix + "embeddings.position_embedding.weight"][:cfg.max_pos]


def arc2face_inverse_face_prompt_embs(sd: SD, cfg: ClipConfig, input_ids: Tensor, face_prompt_embs: Tensor,
                                      pad_embeddings: Tensor, layer_weights=(1.0, 2.0, 4.0), prefix: str = CLIP_PREFIX):
    """arc2face_inverse_face_prompt_embs without extra words (ldm/util.py:1138-1233) on prompt2token_proj: token embeddings of
    "photo of a " + 16 ", " placeholders with positions 4:20 replaced by the core identity embeddings (:1186-1189), CLIP
    forward blending the last len(layer_weights) hidden states (weights normalised to sum 1, arc2face_models.py:230-243),
    core = tokens 4:20, and the 'full_half_pad' variant (:1213-1218: positions 24 .. 24 + (77 - 25) // 2 overwritten by the
    pad embeddings)."""
    tok = clip_embed_tokens(sd, input_ids, prefix).clone()
    tok[:, 4:20] = face_prompt_embs
    full = clip_text_forward(sd, cfg, tok, skip_weights=layer_weights, prefix=prefix)
    half = full.clone()
    pads = full.shape[1] - 25
    if pads >= 2:
        half[:, 24:24 + pads // 2] = pad_embeddings[24:24 + pads // 2]
    return full, half, full[:, 4:20]


def subj_basis_generator_face(sd: SD, cfg: ClipConfig, inverse_ids: Tensor, arc2face_id_embs: Tensor, pad_token: int,
                              out_id_embs_scale: float = 1.0, n_layers: int = 16, layer_weights=(1.0, 2.0, 4.0),
                              prefix: str = CLIP_PREFIX):
    """SubjBasisGenerator.forward, the face / inference branch (subj_basis_generator.py:482-560): core identity embeddings
    [BS, 16, D] from the inverse prompt forward, repeated over the 16 layers (:548-549), blended with pad embeddings 2 .. 2 + K
    by out_id_embs_scale (:553-554).  Returns (static_zs_embs [BS, 16, K, D], 'full_half_pad' inverse prompt embeddings)."""
    pad = clip_pad_embeddings(sd, cfg, pad_token, prefix)
    _, half, core = arc2face_inverse_face_prompt_embs(sd, cfg, inverse_ids, arc2face_id_embs, pad, layer_weights, prefix)
    K = core.shape[1]
    out = core.unsqueeze(1).repeat(1, n_layers, 1, 1) * out_id_embs_scale + pad[2:2 + K].unsqueeze(0) * (1.0 - out_id_embs_scale)
    return out, half
