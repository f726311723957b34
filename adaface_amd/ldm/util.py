"""Hot-path helpers of ldm/util.py: instantiate_from_config (:105-112), get_obj_from_str
(:143-148), load_model_from_config (:114-141).  Configs are plain dicts (or anything with
`in` / [] / .get); yaml is parsed with yaml.safe_load since omegaconf is not installed.
"""
from __future__ import annotations

import importlib

import torch


class _AttrDict(dict):
    """dict with attribute access, enough of OmegaConf's surface for `config.model`."""
    __getattr__ = dict.get

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return _AttrDict({k: _AttrDict.wrap(v) for k, v in obj.items()})
        if isinstance(obj, list):
            return [_AttrDict.wrap(v) for v in obj]
        return obj


def load_config(path):
    import yaml
    with open(path) as f:
        return _AttrDict.wrap(yaml.safe_load(f))


# The reference's yaml names classes under `ldm.`; map the ones on the path onto this package so that
# an unmodified v1-inference-ada.yaml instantiates the HIP-backed modules even when the root-level
# `ldm` shim is not importable.
_TARGET_ALIASES = {
    "ldm.models.diffusion.ddpm.LatentDiffusion": "adaface_amd.ldm.models.diffusion.ddpm.LatentDiffusion",
    "ldm.modules.diffusionmodules.openaimodel.UNetModel": "adaface_amd.ldm.modules.diffusionmodules.openaimodel.UNetModel",
    "ldm.models.autoencoder.AutoencoderKL": "adaface_amd.ldm.models.autoencoder.AutoencoderKL",
    "ldm.modules.embedding_manager.EmbeddingManager": "adaface_amd.ldm.modules.embedding_manager.EmbeddingManager",
    "ldm.modules.encoders.modules.FrozenCLIPEmbedder": "adaface_amd.ldm.modules.encoders.modules.FrozenCLIPEmbedder",
}


def get_obj_from_str(string, reload=False):
    string = _TARGET_ALIASES.get(string, string)
    module, cls = string.rsplit(".", 1)
    if reload:
        importlib.reload(importlib.import_module(module))
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config, **kwargs):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**(config.get("params", None) or dict()), **kwargs)


def load_model_from_config(config, ckpt, verbose=False):
    """ldm/util.py:114-141.  `.ckpt` files are read with weights_only=True (nothing is executed from
    the file); `.safetensors` through safetensors."""
    print(f"Loading model from {ckpt}")
    if ckpt.endswith(".ckpt"):
        pl_sd = torch.load(ckpt, map_location="cpu", weights_only=True)
        sd = pl_sd["state_dict"] if "state_dict" in pl_sd else pl_sd
    elif ckpt.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(ckpt, device="cpu")
    else:
        raise ValueError(f"Unknown checkpoint format: {ckpt}")
    model = instantiate_from_config(config.model if hasattr(config, "model") else config["model"])
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if verbose:
        if missing:
            print("missing keys:", missing)
        if unexpected:
            print("unexpected keys:", unexpected)
    model.eval()
    return model


# ----------------------------------------------------------------------------------------------------------------------
# zero-shot identity path: the two drives of CLIPTextModelWrapper
# ----------------------------------------------------------------------------------------------------------------------
def _tokenize_fixed(tokenizer, prompts, max_length, device, input_ids=None):
    """The reference tokenizes fixed template prompts on every call (util.py:1099-1107, 1174-1180).  There is no CLIP
    vocabulary offline, so `input_ids` ([N, 77] or [1, 77], e.g. saved once with the real tokenizer) may be passed instead."""
    if input_ids is not None:
        return torch.as_tensor(input_ids, dtype=torch.long, device=device)
    if tokenizer is None:
        raise RuntimeError("no tokenizer: pass the CLIP tokenizer or the template's input_ids")
    return tokenizer(prompts, truncation=True, padding="max_length", max_length=max_length,
                     return_tensors="pt").input_ids.to(device)


def arc2face_forward_face_embs(tokenizer, text_encoder, face_embs, input_max_length=77, return_full_and_core_embs=True,
                               input_ids=None, arcface_token_id=None):
    """ldm/util.py:1085-1131.  face_embs [N, 512] normalised ArcFace embeddings -> prompt embeddings of
    "photo of a id person" with the 'id' token's embedding replaced by the zero-padded face vector; full [N, 77, 768] and
    core = tokens 4:20 (the 16 identity embeddings)."""
    import torch.nn.functional as F
    if arcface_token_id is None:
        arcface_token_id = tokenizer.encode("id", add_special_tokens=False)[0]
    ids = _tokenize_fixed(tokenizer, "photo of a id person", input_max_length, face_embs.device, input_ids)
    if ids.shape[0] == 1:
        ids = ids.repeat(len(face_embs), 1)
    hidden = text_encoder.clip_config["hidden"]
    face_embs_padded = F.pad(face_embs.float(), (0, hidden - face_embs.shape[-1]), "constant", 0)
    token_embs = text_encoder(input_ids=ids, return_token_embs=True)
    token_embs[ids == arcface_token_id] = face_embs_padded
    prompt_embeds = text_encoder(input_ids=ids, input_token_embs=token_embs, return_token_embs=False)[0].to(face_embs.dtype)
    if return_full_and_core_embs:
        return prompt_embeds, prompt_embeds[:, 4:20]
    return prompt_embeds[:, 4:20]


def get_b_core_e_embeddings(prompt_embeds, length=22):
    """ldm/util.py:1133-1135."""
    return torch.cat([prompt_embeds[:, :length], prompt_embeds[:, [-1]]], dim=1)


def arc2face_inverse_face_prompt_embs(clip_tokenizer, text_encoder, face_prompt_embs, list_extra_words, return_emb_types,
                                      pad_embeddings, hidden_state_layer_weights=None, input_max_length=77,
                                      zs_extra_words_scale=0.5, input_ids=None):
    """ldm/util.py:1138-1233.  face_prompt_embs [BS, 16, 768] (core identity embeddings) -> prompt2token_proj forward of
    "photo of a " + 16 ", " placeholders [+ extra words] with positions 4:20 replaced, returned in the requested forms."""
    BS = len(face_prompt_embs)
    if list_extra_words is not None:
        if len(list_extra_words) != BS:
            if BS > 1:
                if len(list_extra_words) == 1:
                    list_extra_words = list_extra_words * BS
                else:
                    raise ValueError("list_extra_words has a different length than face_prompt_embs")
            else:
                list_extra_words = list_extra_words[:1]
        for extra_words in list_extra_words:
            assert len(extra_words.split()) <= 2, "Each extra_words string should consist of at most 2 words."
        prompts = ["photo of a " + ", " * 16 + list_extra_words[i] for i in range(len(list_extra_words))]
    else:
        prompts = ["photo of a " + ", " * 16 for _ in range(BS)]
    ids = _tokenize_fixed(clip_tokenizer, prompts, input_max_length, face_prompt_embs.device, input_ids)
    if ids.shape[0] == 1 and BS > 1:
        ids = ids.repeat(BS, 1)
    token_embs = text_encoder(input_ids=ids, return_token_embs=True)
    token_embs[:, 4:20] = face_prompt_embs.float()
    prompt_embeds = text_encoder(input_ids=ids, input_token_embs=token_embs,
                                 hidden_state_layer_weights=hidden_state_layer_weights,
                                 return_token_embs=False)[0].to(face_prompt_embs.dtype)
    core_prompt_embs = prompt_embeds[:, 4:20]
    if list_extra_words is not None:
        core_prompt_embs = torch.cat([core_prompt_embs, prompt_embeds[:, 20:22] * zs_extra_words_scale], dim=1)
    out = []
    for emb_type in return_emb_types:
        if emb_type == "full":
            out.append(prompt_embeds)
        elif emb_type == "full_half_pad":
            pe = prompt_embeds.clone()
            pads = pe.shape[1] - 25
            if pads >= 2:
                pe[:, 24:24 + pads // 2] = pad_embeddings[24:24 + pads // 2]
            out.append(pe)
        elif emb_type == "full_pad":
            pe = prompt_embeds.clone()
            pe[:, 24:-1] = pad_embeddings[24:-1]
            out.append(pe)
        elif emb_type == "core":
            out.append(core_prompt_embs)
        elif emb_type == "full_zeroed_extra":
            pe = prompt_embeds.clone()
            pe[:, 22:24] = pad_embeddings[22:24]
            pe[:, 24:-1] = 0
            out.append(pe)
        elif emb_type == "b_core_e":
            out.append(get_b_core_e_embeddings(prompt_embeds, length=22))
        else:
            raise ValueError(f"unknown emb_type {emb_type}")
    return out
