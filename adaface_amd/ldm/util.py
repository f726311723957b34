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
