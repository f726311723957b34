"""Inference subset of ldm.models.diffusion.ddpm (reference ddpm.py): the pieces of
LatentDiffusion the denoising path touches — model assembly from the yaml config
(ddpm.py:714-813), register_schedule (:244-296), apply_model (:2192-2297),
DiffusionWrapper.forward (:5477-5516), decode_first_stage (:1251-1308), ema_scope (:310-323),
q_sample (:478-482).  ~4.5k lines of training losses are out of scope.

The text side: cond_stage_model = the FrozenCLIPEmbedder drop-in (CLIP text tower on the HIP kernels),
embedding_manager = the inference subset of EmbeddingManager.  get_learned_conditioning takes prompts (needs a tokenizer,
whose vocabulary files do not exist offline), token-id tensors, or a pre-computed static prompt embedding.
"""
from __future__ import annotations

from contextlib import contextmanager
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from adaface_amd.ldm.modules.diffusionmodules.util import extract_into_tensor, make_beta_schedule
from adaface_amd.ldm.util import instantiate_from_config


class DiffusionWrapper(nn.Module):
    """ddpm.py:5477-5516: routes (x, t, c_crossattn=[(emb, prompts, extra_info)]) to the UNet."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key
        if conditioning_key not in (None, "crossattn"):
            raise NotImplementedError(f"conditioning_key '{conditioning_key}' is not used by SD-v1 txt2img")

    def forward(self, x, t, c_concat: list = None, c_crossattn: list = None, cfg_twin: bool = False):
        if self.conditioning_key is None:
            raise NotImplementedError("unconditional UNet is not on the path")
        c0 = c_crossattn[0]
        if isinstance(c0, tuple):
            c_static_emb, c_in, extra_info = c0
        else:
            c_static_emb, c_in, extra_info = c0, None, None
        if cfg_twin:
            return self.diffusion_model(x, t, context=c_static_emb, context_in=c_in, extra_info=extra_info, cfg_twin=True)
        return self.diffusion_model(x, t, context=c_static_emb, context_in=c_in, extra_info=extra_info)


class DDPM(nn.Module):
    """Schedule buffers + UNet wrapper (ddpm.py:73-296), inference arguments only."""

    def __init__(self, unet_config, timesteps=1000, beta_schedule="linear", linear_start=1e-4, linear_end=2e-2,
                 cosine_s=8e-3, given_betas=None, conditioning_key=None, parameterization="eps", use_ema=True,
                 first_stage_key="image", image_size=256, channels=3, log_every_t=100, v_posterior=0.,
                 use_layerwise_embedding=False, **ignored_training_kwargs):
        super().__init__()
        assert parameterization in ("eps", "x0")
        self.parameterization = parameterization
        self.first_stage_key = first_stage_key
        self.image_size = image_size
        self.channels = channels
        self.log_every_t = log_every_t
        self.use_layerwise_embedding = use_layerwise_embedding
        self.N_CA_LAYERS = 16 if use_layerwise_embedding else 1
        self.use_ema = False  # LitEma is training-only; the inference config sets use_ema False (yaml:18)
        self.v_posterior = v_posterior
        self.model = DiffusionWrapper(unet_config, conditioning_key)
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps,
                               linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        """ddpm.py:244-296: numpy fp64 tables stored as fp32 buffers."""
        betas = given_betas if given_betas is not None else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        alphas = 1. - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1., alphas_cumprod[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        to_torch = partial(torch.tensor, dtype=torch.float32)
        self.register_buffer('betas', to_torch(betas))
        self.register_buffer('alphas_cumprod', to_torch(alphas_cumprod))
        self.register_buffer('alphas_cumprod_prev', to_torch(alphas_cumprod_prev))
        self.register_buffer('sqrt_alphas_cumprod', to_torch(np.sqrt(alphas_cumprod)))
        self.register_buffer('sqrt_one_minus_alphas_cumprod', to_torch(np.sqrt(1. - alphas_cumprod)))
        self.register_buffer('sqrt_recip_alphas_cumprod', to_torch(np.sqrt(1. / alphas_cumprod)))
        self.register_buffer('sqrt_recipm1_alphas_cumprod', to_torch(np.sqrt(1. / alphas_cumprod - 1)))

    @property
    def device(self):
        return self.betas.device

    @contextmanager
    def ema_scope(self, context=None):
        yield None  # use_ema is False at inference (ddpm.py:310-323 is then a no-op)

    def q_sample(self, x_start, t, noise=None):
        """ddpm.py:478-482."""
        noise = torch.randn_like(x_start) if noise is None else noise
        return (extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start +
                extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)


class LatentDiffusion(DDPM):
    """ddpm.py:712-5396, inference subset."""

    def __init__(self, first_stage_config, cond_stage_config=None, personalization_config=None,
                 num_timesteps_cond=None, cond_stage_key="image", cond_stage_trainable=False, concat_mode=True,
                 cond_stage_forward=None, conditioning_key=None, scale_factor=1.0, scale_by_std=False,
                 *args, **kwargs):
        if conditioning_key is None:
            conditioning_key = 'concat' if concat_mode else 'crossattn'
        unet_config = kwargs.pop("unet_config")
        kwargs.pop("ckpt_path", None)
        kwargs.pop("ignore_keys", None)
        super().__init__(unet_config, conditioning_key=conditioning_key, *args, **kwargs)
        self.cond_stage_key = cond_stage_key
        self.cond_stage_trainable = cond_stage_trainable
        self.scale_factor = scale_factor
        self.first_stage_model = instantiate_from_config(first_stage_config).eval()
        # ddpm.py:808-813, 860-880: the conditioning producer.  Constructed like the reference does so that the
        # `cond_stage_model.*` entries of an SD checkpoint load; its HIP engine is only created on first use.
        self.cond_stage_config = cond_stage_config
        self.personalization_config = personalization_config
        self.cond_stage_forward = cond_stage_forward
        self.cond_stage_model = None
        if isinstance(cond_stage_config, dict) and "target" in cond_stage_config:
            self.cond_stage_model = instantiate_from_config(cond_stage_config).eval()
        self.embedding_manager = None
        if self.cond_stage_model is not None:
            from adaface_amd.ldm.modules.embedding_manager import EmbeddingManager
            if isinstance(personalization_config, dict) and "target" in personalization_config:
                pc = dict(personalization_config.get("params", None) or {})
                pc.pop("embedding_manager_ckpt", None)
                self.embedding_manager = EmbeddingManager(self.cond_stage_model, **pc)
            else:
                self.embedding_manager = EmbeddingManager(self.cond_stage_model, subject_strings=[])
        self.compel_cfg_weight_level_range = None
        self.apply_compel_cfg_prob = 0
        self.empty_context = None

    # ---- conditioning ------------------------------------------------------------------------
    def get_learned_conditioning(self, cond_in, zs_clip_features=None, zs_id_embs=None,
                                 zs_out_id_embs_scale_range=(1.0, 1.0), randomize_clip_weights=False,
                                 apply_arc2face_inverse_embs=False, apply_arc2face_embs=False, embman_iter_type=None):
        """ddpm.py:962-1076.  Accepts a pre-computed static prompt embedding tensor [B*16,77,768] (or
        [B,77,768], replicated over the 16 layers like embedding_manager.py:1342-1353) and wraps it into the
        (emb, prompts, extra_info) tuple the sampler and the UNet expect (ddpm.py:1056-1069)."""
        if isinstance(cond_in, torch.Tensor) and cond_in.is_floating_point():
            emb = cond_in
            if self.use_layerwise_embedding and emb.dim() == 3 and emb.shape[0] % self.N_CA_LAYERS != 0:
                raise ValueError("layerwise embedding must have batch divisible by 16")
            prompts = [""] * (emb.shape[0] // (self.N_CA_LAYERS if self.use_layerwise_embedding else 1))
            extra_info = {'use_layerwise_context': self.use_layerwise_embedding, 'use_conv_attn_kernel_size': -1,
                          'placeholder2indices': None, 'prompt_emb_mask': None, 'is_training': False,
                          'compel_cfg_weight_level_range': None, 'apply_compel_cfg_prob': 0,
                          'empty_context': self.empty_context, 'capture_distill_attn': False}
            return (emb, prompts, extra_info)
        # prompts (list of str) or token ids (int64 [B, 77]): the reference's flow, ddpm.py:966-1069
        if apply_arc2face_inverse_embs or apply_arc2face_embs:
            raise NotImplementedError("apply_arc2face_(inverse_)embs replace the prompt embeddings in training-time iterations "
                                      "(ddpm.py:1010-1053): not part of the inference path")
        if zs_clip_features is not None or zs_id_embs is not None:          # ddpm.py:992-999
            if not getattr(self.embedding_manager, "do_zero_shot", False):
                raise NotImplementedError("zero-shot identity conditioning needs an EmbeddingManager built with do_zero_shot=True "
                                          "(plus the Arc2Face encoder and SubjBasisGenerator weights, absent offline)")
            self.embedding_manager.set_zs_image_features(zs_clip_features, zs_id_embs, zs_out_id_embs_scale_range)
        if self.cond_stage_model is None:
            raise RuntimeError("this LatentDiffusion was built without a cond_stage_config: pass an embedding tensor")
        self.cond_stage_model.device = self.device
        if self.empty_context is not None:
            self.empty_context = self.empty_context.to(self.device)
        if randomize_clip_weights:
            self.cond_stage_model.sample_last_layers_skip_weights()
        self.embedding_manager.set_curr_iter_type(embman_iter_type or 'recon_iter')
        static_prompt_embedding = self.cond_stage_model.encode(cond_in, embedding_manager=self.embedding_manager)
        import copy
        extra_info = {'use_layerwise_context': self.use_layerwise_embedding,
                      'use_conv_attn_kernel_size': self.embedding_manager.use_conv_attn_kernel_size,
                      'placeholder2indices': copy.copy(self.embedding_manager.placeholder2indices),
                      'prompt_emb_mask': copy.copy(self.embedding_manager.prompt_emb_mask),
                      'is_training': self.embedding_manager.training,
                      'compel_cfg_weight_level_range': self.compel_cfg_weight_level_range,
                      'apply_compel_cfg_prob': self.apply_compel_cfg_prob,
                      'empty_context': self.empty_context,
                      'capture_distill_attn': False}
        prompts = list(cond_in) if not isinstance(cond_in, torch.Tensor) else [""] * cond_in.shape[0]
        return (static_prompt_embedding, prompts, extra_info)

    @staticmethod
    def layerwise_repeat(emb, n_layers=16):
        """[B,T,D] -> [B*16,T,D] with the layer copies adjacent per instance (embedding_manager.py:1342-1353)."""
        B, T, D = emb.shape
        return emb.unsqueeze(1).expand(B, n_layers, T, D).reshape(B * n_layers, T, D).contiguous()

    # ---- denoiser ----------------------------------------------------------------------------
    def apply_model(self, x_noisy, t, cond, return_ids=False):
        """ddpm.py:2192-2297 (the split_input_params patch mode is never active: SURVEY.md §8a a6)."""
        if not isinstance(cond, dict):
            if not isinstance(cond, list):
                cond = [cond]
            key = 'c_concat' if self.model.conditioning_key == 'concat' else 'c_crossattn'
            cond = {key: cond}
        x_recon = self.model(x_noisy, t, **cond)
        if isinstance(x_recon, tuple) and not return_ids:
            return x_recon[0]
        return x_recon

    def apply_model_cfg_twin(self, x_noisy, t, cond_twin):
        """apply_model(torch.cat([x] * 2), torch.cat([t] * 2), cond_twin) -- the classifier-free-guidance call of
        p_sample_ddim / p_sample_plms (ddim.py:236-247) -- without the concatenation: `cond_twin` is the condition of the
        2B samples (cond first), x_noisy / t those of one half.  Returns eps of the 2B samples.  Not a reference method: the
        drop-in samplers use it when the model has it, and fall back to apply_model on the concatenated batch otherwise."""
        cond = cond_twin
        if not isinstance(cond, dict):
            if not isinstance(cond, list):
                cond = [cond]
            cond = {'c_crossattn': cond}
        if 'c_crossattn' not in cond:
            return self.apply_model(torch.cat([x_noisy] * 2), torch.cat([t] * 2), cond_twin)
        return self.model(x_noisy, t, cfg_twin=True, **cond)

    # ---- decoder -----------------------------------------------------------------------------
    @torch.no_grad()
    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        """ddpm.py:1251-1308: z / scale_factor -> first_stage_model.decode; the division is folded into
        the NCHW->NHWC conversion kernel at the head of the HIP decoder."""
        if predict_cids:
            raise NotImplementedError("VQ codebook ids are not used by the KL autoencoder")
        return self.first_stage_model.decode(z, scale_factor=self.scale_factor)

    @torch.no_grad()
    def decode_first_stage_uint8(self, z):
        """decode + clamp((x+1)/2,0,1)*255 -> uint8 HWC (stable_txt2img.py:713-715,764-765) in one pass."""
        return self.first_stage_model.decode(z, scale_factor=self.scale_factor, return_uint8=True)

    @torch.no_grad()
    def encode_first_stage(self, x, mask=None):
        """ddpm.py:1372-1410 (the non-patched branch): first_stage_model.encode -> posterior."""
        return self.first_stage_model.encode(x, mask)

    @torch.no_grad()
    def get_first_stage_encoding(self, encoder_posterior):
        """ddpm.py:947-954: sample the posterior (a tensor passes through) and apply scale_factor."""
        from adaface_amd.ldm.models.autoencoder import DiagonalGaussianDistribution
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            return encoder_posterior.sample(scale=self.scale_factor)
        if isinstance(encoder_posterior, torch.Tensor):
            return self.scale_factor * encoder_posterior
        raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")

    def set_compute_dtype(self, dtype: str):
        """'bf16' | 'f32' | 'fp8' (the UNet's ResBlock convolutions in e4m3; VAE and text tower stay bf16)."""
        rest = "bf16" if dtype == "fp8" else dtype
        self.model.diffusion_model.set_compute_dtype(dtype)
        self.first_stage_model.set_compute_dtype(rest)
        if self.cond_stage_model is not None:
            self.cond_stage_model.set_compute_dtype(rest)
        return self
