"""The model section of configs/stable-diffusion/v1-inference-ada.yaml (reference :1-76) as a plain
dict — the hyper-parameters of the denoising path (SURVEY.md §8 'config constants')."""
from __future__ import annotations

import copy


def sd15_config(use_layerwise_embedding: bool = True) -> dict:
    return copy.deepcopy({
        "model": {
            "target": "ldm.models.diffusion.ddpm.LatentDiffusion",
            "params": {
                "linear_start": 0.00085, "linear_end": 0.0120, "num_timesteps_cond": 1, "log_every_t": 200,
                "timesteps": 1000, "first_stage_key": "jpg", "cond_stage_key": "txt", "image_size": 64,
                "channels": 4, "cond_stage_trainable": False, "conditioning_key": "crossattn",
                "scale_factor": 0.18215, "use_ema": False, "use_layerwise_embedding": use_layerwise_embedding,
                "unet_config": {
                    "target": "ldm.modules.diffusionmodules.openaimodel.UNetModel",
                    "params": {"image_size": 32, "in_channels": 4, "out_channels": 4, "model_channels": 320,
                               "attention_resolutions": [4, 2, 1], "num_res_blocks": 2, "channel_mult": [1, 2, 4, 4],
                               "num_heads": 8, "use_spatial_transformer": True, "transformer_depth": 1,
                               "context_dim": 768, "use_checkpoint": True, "legacy": False},
                },
                "first_stage_config": {
                    "target": "ldm.models.autoencoder.AutoencoderKL",
                    "params": {"embed_dim": 4, "monitor": "val/rec_loss",
                               "ddconfig": {"double_z": True, "z_channels": 4, "resolution": 256, "in_channels": 3,
                                            "out_ch": 3, "ch": 128, "ch_mult": [1, 2, 4, 4], "num_res_blocks": 2,
                                            "attn_resolutions": [], "dropout": 0.0},
                               "lossconfig": {"target": "torch.nn.Identity"}},
                },
                "cond_stage_config": {"target": "ldm.modules.encoders.modules.FrozenCLIPEmbedder"},
            },
        }
    })


def tiny_config() -> dict:
    """Same block / cross-attention layout as SD-1.5, 5x narrower (tests and smoke)."""
    cfg = sd15_config()
    p = cfg["model"]["params"]
    p["unet_config"]["params"].update(model_channels=64, num_heads=2, context_dim=64)
    p["first_stage_config"]["params"]["ddconfig"].update(ch=64, ch_mult=[1, 2, 2, 2])
    # a text tower of the same structure as CLIP ViT-L/14's, 12x narrower, whose width is the tiny U-Net's context_dim
    p["cond_stage_config"]["params"] = {"clip_config": dict(vocab=1000, hidden=64, layers=3, heads=4, intermediate=128,
                                                            max_pos=77)}
    return cfg
